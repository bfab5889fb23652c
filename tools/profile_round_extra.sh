# Round-6 extras behind tools/profile_round.sh (run on the GPU box from the repo root): the secondary
# operating point, the reader-fed lines, the same-device multi-rank lines with the per-rank fields,
# host issue time, the RCCL rehearsal, the loss curve.  Output: gpurun_out/${T}_summaries/.
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${PROFILE_TAG:-r06}
S=$R/gpurun_out/${T}_summaries
mkdir -p $S
cd $R
for CFG in c1 c2; do
  timeout 300 python bench.py --config $CFG --no-cpu-baseline --image-hw 1000 1333 --batch 2 --proposals 500 > $S/${T}_bench_${CFG}_1000px.json 2> $S/err_${CFG}_1000px.txt
  timeout 300 python bench.py --config $CFG --no-cpu-baseline --no-plan > $S/${T}_bench_${CFG}_python_driven.json 2> $S/err_${CFG}_noplan.txt
  timeout 600 python bench.py --reader --config $CFG --steps 300 --warmup 20 > $S/${T}_bench_reader_${CFG}.json 2> $S/err_reader_${CFG}.txt
  timeout 600 python bench.py --reader --config $CFG --steps 300 --warmup 20 --image-hw 1000 1333 --batch 2 --proposals 500 > $S/${T}_bench_reader_${CFG}_1000px.json 2> $S/err_reader_${CFG}_1000px.txt
done
timeout 600 python bench.py --reader --config c2 --steps 300 --warmup 20 --image-hw 1000 1333 --batch 2 --proposals 500 --no-plan > $S/${T}_bench_reader_c2_1000px_python_driven.json 2> $S/err_reader_c2_1000px_noplan.txt
timeout 300 python bench.py --no-f32x9 --no-cpu-baseline > $S/${T}_bench_c1_fp32_mfma.json 2> $S/err_no_f32x9.txt
timeout 300 python bench.py --config c3 --no-f32x9 --no-cpu-baseline > $S/${T}_bench_c3_fp32_mfma.json 2> $S/err_no_f32x9_c3.txt
timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline --available-cus 224 2> $S/err_gpus8.txt | grep '^{' | tail -1 > $S/${T}_bench_gpus8_same_device.json
timeout 300 python tools/host_time.py > $S/${T}_host_issue_time.txt 2>&1
timeout 600 python tools/rccl_rehearsal.py 2> $S/err_rccl.txt | grep '^{' | tail -1 > $S/${T}_rccl_rehearsal.json
timeout 900 python tools/loss_curve.py $S/${T}_loss_curve.json --steps 400 > $S/loss_curve.log 2>&1
ls -la $S
