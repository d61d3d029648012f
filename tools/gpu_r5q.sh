#!/bin/bash
# (GPU box) LatticeFold+ set check with rounds 0-1 from the exponent digits: protocol tests, stage timeline, proves
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_scale.py tests/test_gpu_lfplus_prover.py tests/test_gpu_calibration.py tests/test_dist_shard.py::test_model_transport_runs_a_ranks_share_and_counts_its_exchanges -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r5q_lfp_tests.txt
LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 --resident > gpurun_out/r5q_lfplus_p20_full.txt 2>&1
grep -v "round " gpurun_out/r5q_lfplus_p20_full.txt | tail -26 > gpurun_out/r5q_lfplus_p20.txt; cat gpurun_out/r5q_lfplus_p20.txt
for nv in 15 17 20; do timeout 600 python tools/bench_lfplus.py --nvars $nv --k 4 --fresh 3 --rounds 3 --resident 2>/dev/null | tail -1; done | tee gpurun_out/r5q_lfplus_ms.txt
LFPLUS_CM_DENSE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 3 --resident 2>/dev/null | tail -1 | tee -a gpurun_out/r5q_lfplus_ms.txt
