O=gpurun_out/r2k; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|GRBM_GUI_ACTIVE|SQ_BUSY_CU|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAVES" | head -40 > $O/counters.txt; cat $O/counters.txt | cut -c1-200
export C2D_TUNE=1
for sp in 1 2 3 4; do echo "== crop split $sp"; C2D_CROP_SPLIT=$sp timeout 100 python tools/bench_crop_fwd.py 2>&1 | grep -v amdgpu; done
