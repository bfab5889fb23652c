set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
C2D_TUNE=1 C2D_CROP_STREAM=2 python -m pytest tests/test_gpu_ops.py tests/test_golden_vectors.py -k "crop or roi" -x -q 2>&1 | tail -5 > $O/r4_tests13.log
echo form1; python tools/bench_crop_fwd.py
echo form2; C2D_TUNE=1 C2D_CROP_STREAM=2 python tools/bench_crop_fwd.py
for SP in 2 3 4 6 8; do echo form2 split $SP; C2D_TUNE=1 C2D_CROP_STREAM=2 C2D_CROP_SPLIT=$SP python tools/bench_crop_fwd.py; done
tail -3 $O/r4_tests13.log
