#!/bin/bash
# (GPU box) paired vs unpaired digit-plane commits: parity test, then C4 step time / commit-kernel time for both, rocprof kernel stats of the paired run
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python -m pytest $R/tests/test_gpu_ajtai_i8.py::test_paired_commit_of_both_decompositions -x -q 2>&1 | tail -5
for v in "" 1; do
  if [ -z "$v" ]; then unset LF_I8_PAIR; else export LF_I8_PAIR=1; fi
  python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels']['k_ajtai_i8']
print('PAIR=$v', 'ms/step', round(d['ms_per_step'], 3), 'commit kernel avg ms', round(k['avg_ms'], 3), 'launches/step', k['launches_per_step'], 'hbm_8d frac', round(d['roofline']['hbm_8d']['frac'], 3), 'mfma frac', round(d['roofline']['frac'], 3), {a: round(b, 2) for a, b in d['phases_ms_per_step'].items()})"
done
unset LF_I8_PAIR
