#!/bin/bash
# (GPU box) commit-kernel time at C4 over the measurement switches of lf_ajtai_i8.hip: usage tools/gpu_i8sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "$@"; do
  env $cfg python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels']['k_ajtai_i8']
print('%-40s' % '$cfg', 'ms/step', round(d['ms_per_step'], 3), 'commit kernel avg ms', round(k['avg_ms'], 3), 'x', k['launches_per_step'])"
done
