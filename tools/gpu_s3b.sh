#!/bin/bash
# (GPU box) from_f inside the step: tests of both forms, C4 / C3 lines with and without it, and the two-stream throughput mode
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bb.py -q -m gpu -x -k "builds_f_and_w_ccs or test_fold_step_parity or witness_roundtrips" 2>&1 | tail -4 > gpurun_out/s3b_tests.txt
cat gpurun_out/s3b_tests.txt
for rep in 1 2; do
  for lazy in 0 1; do
    if [ $lazy = 1 ]; then export LF_LAZY_FROM_F=1; else unset LF_LAZY_FROM_F; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 lazy=$lazy', d['ms_per_step'])" | tee -a gpurun_out/s3b_ab.txt
    python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 lazy=$lazy', d['ms_per_step'])" | tee -a gpurun_out/s3b_ab.txt
  done
done
unset LF_LAZY_FROM_F
python bench.py --steps 10 --warmup 2 --streams 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 streams=2', d['ms_per_step'], d['value'])" | tee -a gpurun_out/s3b_ab.txt
python bench.py --steps 10 --warmup 2 --streams 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 streams=3', d['ms_per_step'], d['value'])" | tee -a gpurun_out/s3b_ab.txt
for r in goldilocks babybear frog; do python tools/time_poseidon.py $r; done | tee gpurun_out/s3b_poseidon.txt
lscpu | grep -E "Model name|MHz|^CPU\(s\)" | tee -a gpurun_out/s3b_poseidon.txt
