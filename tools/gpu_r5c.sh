#!/bin/bash
# (GPU box) prefetch, second pass: tests, then same-box A/B over the early trigger points and stream priorities; sharded tests (new schedule / hand-over)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prefetch.py tests/test_gpu_lfplus_prover.py::test_scratch_cache_is_bounded_and_released -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r5c_tests.txt
cat gpurun_out/r5c_tests.txt
b() { python bench.py --no-cpu-baseline --no-lfplus "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('matches_oracle_fixture'), (d['config'].get('prefetch') or {}).get('used'))"; }
for rep in 1 2; do
echo "C4 no-prefetch: $(b --no-prefetch)"
for at in 0 1 2; do echo "C4 LF_PF_AT=$at: $(LF_PF_AT=$at b)"; done
for at in 0 1; do echo "C4 LF_PF_AT=$at LANE0_MID: $(LF_LANE0_MID=1 LF_PF_AT=$at b)"; done
echo "C4 no-prefetch LANE0_MID: $(LF_LANE0_MID=1 b --no-prefetch)"
done
echo "C2 no-prefetch: $(b --workload C2 --steps 20 --warmup 3 --no-prefetch)"
for at in 0 1 2; do echo "C2 LF_PF_AT=$at: $(LF_PF_AT=$at b --workload C2 --steps 20 --warmup 3)"; done
for at in 0 1; do
LF_PF_AT=$at LF_TIMELINE=1 python bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[timeline\]" | tail -36 > gpurun_out/r5c_timeline_c4_pf$at.txt
done
timeout 1500 python -m pytest tests/test_dist_shard.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/r5c_shard_tests.txt
cat gpurun_out/r5c_shard_tests.txt
