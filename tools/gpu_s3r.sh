#!/bin/bash
# (GPU box) after the host-Poseidon work: bench lines (same box: old Poseidon via LF_POSEIDON_SCALAR is not comparable, so only absolute numbers), LF+ P20 / P17, LF+ tests
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
bash tools/gpu_quick.sh s3r
python tools/bench_lfplus.py --nvars 17 20 --k 4 --fresh 3 --rounds 3 --resident 2>&1 | tail -4 | tee gpurun_out/s3r_lfplus.txt
timeout 900 python -m pytest tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_prover.py tests/test_gpu_bb.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/s3r_tests.txt
LF_TIMELINE=1 python bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[timeline\]" | tail -32 > gpurun_out/s3r_timeline_c4.txt
LF_TIMELINE=1 python bench.py --workload C3 --steps 1 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[bb timeline\]" | tail -36 > gpurun_out/s3r_timeline_c3.txt
