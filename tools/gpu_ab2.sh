#!/bin/bash
# (GPU box) A/B of an environment assignment: tools/gpu_ab2.sh VAR VALUE [workload] -- alternates `VAR unset` / `VAR=VALUE` three times
R=${GRAFT_REPO_ROOT:-$(pwd)}
var=$1; val=$2; wl=${3:-C4}
for i in 1 2 3; do
  for v in "" "$val"; do
    if [ -z "$v" ]; then unset $var; else export $var=$v; fi
    python $R/bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['ms_per_step'], 3), round(d['phases_ms_per_step']['fold_sumcheck'], 2))"
  done
done
