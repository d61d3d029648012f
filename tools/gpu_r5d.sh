#!/bin/bash
# (GPU box) prefetch in two parts (LF_PF_AT: bit planes + z_k, LF_PF_AT2: commits): tests, then same-box A/B of the C4 step over the trigger points
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prefetch.py -q -m gpu 2>&1 | tail -12 > gpurun_out/r5d_tests.txt
cat gpurun_out/r5d_tests.txt
b() { python bench.py --no-cpu-baseline --no-lfplus "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('matches_oracle_fixture'), (d['config'].get('prefetch') or {}).get('used'))"; }
for rep in 1 2; do
echo "C4 no-prefetch: $(b --no-prefetch)"
for cfg in "0 0" "2 2" "2 16" "2 17" "2 18" "2 40" "0 17" "16 16" "17 17" "18 18" "17 40" "3 17" "2 50" "40 50"; do set -- $cfg; echo "C4 LF_PF_AT=$1 AT2=$2: $(LF_PF_AT=$1 LF_PF_AT2=$2 b)"; done
done
for cfg in "2 17" "17 17" "2 40"; do set -- $cfg
LF_PF_AT=$1 LF_PF_AT2=$2 LF_TIMELINE=1 python bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[timeline\]" | tail -38 > gpurun_out/r5d_timeline_c4_pf$1_$2.txt
done
