# Counter calibration (run on the GPU box from the repo root: `bash tools/calib/run_calibration.sh`).
# Four SEPARATE rocprofv3 passes over the known-answer program tools/calib/counter_calib (built
# in-tree by `hipcc -O3 --offload-arch=gfx950 -o tools/calib/counter_calib tools/calib/counter_calib.hip`;
# the program directly after `--`, PMC passes with --kernel-trace only), reduced by
# tools/calib/summarize_calibration.py into profiles/rNN_counter_calibration.json.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=${PROFILE_TAG:-r04}
B=$R/tools/calib/counter_calib
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
$B > $O/${T}_calib_plain.json 2> $O/${T}_calib_plain.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_calib_stats -o s -- $B > $O/${T}_calib_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_calib_fetch -o f -- $B > $O/${T}_calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_calib_write -o w -- $B > $O/${T}_calib_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${T}_calib_mfma -o m -- $B > $O/${T}_calib_mfma.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/${T}_calib_tcc -o t -- $B > $O/${T}_calib_tcc.log 2>&1
cd $R
python3 tools/calib/summarize_calibration.py $O/${T}_counter_calibration.json $O/${T}_calib_plain.json $O/${T}_calib_fetch $O/${T}_calib_write $O/${T}_calib_mfma $O/${T}_calib_stats $O/${T}_calib_tcc
find $O -name "*kernel_trace.csv" -path "*${T}_calib*" -delete
