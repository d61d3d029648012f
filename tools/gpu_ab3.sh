#!/bin/bash
# (GPU box) A/B/C of three builds of the library on one workload: tools/gpu_ab3.sh a.so b.so c.so [workload]
R=${GRAFT_REPO_ROOT:-$(pwd)}
wl=${4:-C3}
cp $R/latticefold_amd/liblfhip.so /tmp/keep.so
for i in 1 2 3; do
  for v in $1 $2 $3; do
    cp $R/$v $R/latticefold_amd/liblfhip.so
    python $R/bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'], 3), round(d['phases_ms_per_step'].get('fold_finish', 0), 2))"
  done
done
cp /tmp/keep.so $R/latticefold_amd/liblfhip.so
