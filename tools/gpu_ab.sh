#!/bin/bash
# (GPU box) A/B of an environment switch: tools/gpu_ab.sh VAR [workload] -- alternates `VAR unset` / `VAR=1` three times
R=${GRAFT_REPO_ROOT:-$(pwd)}
var=$1; wl=${2:-C4}
for i in 1 2 3; do
  for v in "" 1; do
    if [ -z "$v" ]; then unset $var; else export $var=$v; fi
    python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['ms_per_step'], 3), round(d['phases_ms_per_step']['fold_sumcheck'], 2))"
  done
done
