#!/bin/bash
# (GPU box) interleaved A/B of two builds of the library on the bench step: tools/gpu_ablib.sh a.so b.so [workload]   (paths relative to the repo root; either may
# be the installed latticefold_amd/liblfhip.so -- both are copied aside first)
R=${GRAFT_REPO_ROOT:-$(pwd)}
wl=${3:-C4}
cp $R/$1 /tmp/ab_a.so; cp $R/$2 /tmp/ab_b.so; cp $R/latticefold_amd/liblfhip.so /tmp/keep.so
for i in 1 2 3; do
  for v in a b; do
    cp /tmp/ab_$v.so $R/latticefold_amd/liblfhip.so
    python $R/bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus --no-ajtai --no-shard-model --chain 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels']
aj = [v for n, v in k.items() if n.startswith('k_ajtai')]
print('$v', round(d['ms_per_step'], 3), 'commit launch ms', round(aj[0]['avg_ms'], 4) if aj else None, {n: round(x, 2) for n, x in d['phases_ms_per_step'].items() if n in ('linearization', 'decomp_crt_commit', 'fold_sumcheck')})"
  done
done
cp /tmp/keep.so $R/latticefold_amd/liblfhip.so
