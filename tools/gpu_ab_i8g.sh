#!/bin/bash
# (GPU box) interleaved A/B of the general commit kernel in two builds of the library: tools/gpu_ab_i8g.sh old.so new.so  (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cp $R/$1 /tmp/ab_a.so; cp $R/$2 /tmp/ab_b.so; cp $R/latticefold_amd/liblfhip.so /tmp/keep.so
for i in 1 2 3; do
  for v in a b; do
    cp /tmp/ab_$v.so $R/latticefold_amd/liblfhip.so
    echo "== $v"
    timeout 200 python $R/tools/time_commit_general.py ${3:-C4} 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('commit_ntt', [round(x, 3) for x in d['commit_ntt_kernel_ms'][1:]], 'witness', [round(x[0], 3) for x in d['witness_commit_kernel_ms_wall_ms'][1:]])"
  done
done
cp /tmp/keep.so $R/latticefold_amd/liblfhip.so
