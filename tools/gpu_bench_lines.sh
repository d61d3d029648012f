#!/bin/bash
# (GPU box) the three committed bench lines of a round from the current build and the committed profiles: usage tools/gpu_bench_lines.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r03c}
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/${tag}_bench_c4.json 2> $R/gpurun_out/${tag}_bench_c4.err
python $R/bench.py --workload C2 --steps 20 --warmup 3 > $R/gpurun_out/${tag}_bench_c2.json 2>/dev/null
python $R/bench.py --workload C3 --steps 5 --warmup 2 > $R/gpurun_out/${tag}_bench_c3.json 2>/dev/null
for w in c4 c2 c3; do python -c "
import json;d=json.load(open('$R/gpurun_out/${tag}_bench_$w.json'));print('$w', round(d['value'],2), round(d['ms_per_step'],3), d['roofline'].get('traffic_source','')[:40])"; done
