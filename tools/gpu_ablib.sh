#!/bin/bash
# (GPU box) A/B of two builds of the library: tools/gpu_ablib.sh old.so new.so [workload]
R=${GRAFT_REPO_ROOT:-$(pwd)}
wl=${3:-C4}
cp $R/latticefold_amd/liblfhip.so /tmp/keep.so
for i in 1 2 3; do
  for v in $1 $2; do
    cp $R/$v $R/latticefold_amd/liblfhip.so
    python $R/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'], 3), {k: round(x, 2) for k, x in d['phases_ms_per_step'].items() if k in ('linearization', 'decomp_crt_commit', 'fold_sumcheck')})"
  done
done
cp /tmp/keep.so $R/latticefold_amd/liblfhip.so
