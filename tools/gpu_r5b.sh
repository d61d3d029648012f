#!/bin/bash
# (GPU box) prefetch: tests, then same-box A/B of the C4 / C2 step with and without the hint and over the trigger points
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prefetch.py tests/test_gpu_lfplus_prover.py::test_scratch_cache_is_bounded_and_released -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r5b_tests.txt
cat gpurun_out/r5b_tests.txt
b() { python bench.py --no-cpu-baseline --no-lfplus "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('matches_oracle_fixture'), (d['config'].get('prefetch') or {}).get('used'))"; }
for rep in 1 2; do
echo "C4 no-prefetch: $(b --no-prefetch)"
for at in 0 1 13 14 15 16 17 18 2 3; do echo "C4 LF_PF_AT=$at: $(LF_PF_AT=$at b)"; done
done
echo "C2 no-prefetch: $(b --workload C2 --steps 20 --warmup 3 --no-prefetch)"
for at in 0 1 16 2 3; do echo "C2 LF_PF_AT=$at: $(LF_PF_AT=$at b --workload C2 --steps 20 --warmup 3)"; done
LF_TIMELINE=1 python bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[timeline\]" | tail -45 > gpurun_out/r5b_timeline_c4_pf.txt
