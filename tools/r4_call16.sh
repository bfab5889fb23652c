set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
for i in a b; do
  timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/r4k_bench_c2_$i.json 2> /dev/null
  C2D_FUSE_BN_BWD=1 timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/r4k_bench_c2_fusebn_$i.json 2> /dev/null
done
timeout 300 python bench.py --no-cpu-baseline --image-hw 1000 1333 --batch 2 --proposals 500 > $O/r04_bench_c1_1000px.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --dtype bf16 --image-hw 1000 1333 --batch 2 --proposals 500 > $O/r04_bench_c2_1000px.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --graph > $O/r04_bench_c1_graph.json 2> /dev/null
timeout 300 python bench.py --config c2 --no-cpu-baseline --graph > $O/r04_bench_c2_graph.json 2> /dev/null
C2D_GRAPH_STREAMS=0 timeout 300 python bench.py --no-cpu-baseline --graph > $O/r04_bench_c1_graph_one_stream.json 2> /dev/null
C2D_GRAPH_STREAMS=0 timeout 300 python bench.py --config c2 --no-cpu-baseline --graph > $O/r04_bench_c2_graph_one_stream.json 2> /dev/null
for f in r4k_bench_c2_a r4k_bench_c2_fusebn_a r4k_bench_c2_b r4k_bench_c2_fusebn_b r04_bench_c1_1000px r04_bench_c2_1000px r04_bench_c1_graph r04_bench_c2_graph r04_bench_c1_graph_one_stream r04_bench_c2_graph_one_stream; do python3 -c "
import json
l=[x for x in open('$O/$f.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$f', d['ms_per_step'] if d else 'NO LINE', d['value'] if d else '')"; done
