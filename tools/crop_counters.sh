#!/bin/bash
# PMC passes for the two ROI-crop kernels (forward stream kernel, backward strip kernel) on the
# benchmark's map / box distribution (tools/bench_crop.py, fp32; C2D_CROP_BF16=1: bf16 output /
# gradient).  Separate runs per counter set, --kernel-trace only beside --pmc, the program directly
# after `--` (GPU box, repo root).  Output: gpurun_out/crop_pmc/<set>/  -> tools/summarize_crop_counters.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/crop_pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o c -- python3 $R/tools/bench_crop.py > $O/$name.log 2>&1 || echo "pass $name failed" ; }
run sq_time   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq_insts  SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE
run tcp       TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcp_stall TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tcc       TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr
run fetch     FETCH_SIZE
run write     WRITE_SIZE
python3 $R/tools/summarize_crop_counters.py $O $O/crop_counters.json
cat $O/crop_counters.json | head -120
