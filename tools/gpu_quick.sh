#!/bin/bash
# (GPU box) quick A/B: bench lines for C4 / C2 / C3 (no cpu baseline), tag = $1; extra env passes through
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-q}
mkdir -p $R/gpurun_out
python $R/bench.py --no-cpu-baseline --no-lfplus > $R/gpurun_out/${tag}_c4.json 2> $R/gpurun_out/${tag}_c4.err
python $R/bench.py --workload C2 --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus > $R/gpurun_out/${tag}_c2.json 2>/dev/null
python $R/bench.py --workload C3 --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus > $R/gpurun_out/${tag}_c3.json 2>/dev/null
cat $R/gpurun_out/${tag}_c4.json $R/gpurun_out/${tag}_c2.json $R/gpurun_out/${tag}_c3.json | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['config']['workload'], d['ms_per_step'], d['value'])
"
