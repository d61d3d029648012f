#!/bin/bash
# (GPU box) LatticeFold+ with the IFMA lanes of the Frog transcript: host permutation times, the P20 stage timeline with and without them, digests at scale
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
( python tools/time_poseidon.py frog; LFPLUS_POSEIDON_SCALAR=1 python tools/time_poseidon.py frog; python tools/time_poseidon.py goldilocks; lscpu | grep "Model name" ) > gpurun_out/r5i_poseidon.txt 2>&1
cat gpurun_out/r5i_poseidon.txt
LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 --resident > gpurun_out/r5i_lfplus_p20_full.txt 2>&1
grep -v "round " gpurun_out/r5i_lfplus_p20_full.txt | tail -26 > gpurun_out/r5i_lfplus_p20.txt; cat gpurun_out/r5i_lfplus_p20.txt
for nv in 15 17 20; do timeout 600 python tools/bench_lfplus.py --nvars $nv --k 4 --fresh 3 --rounds 3 --resident 2>/dev/null | tail -1; done | tee gpurun_out/r5i_lfplus_ms.txt
LFPLUS_POSEIDON_SCALAR=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 3 --resident 2>/dev/null | tail -1 | tee -a gpurun_out/r5i_lfplus_ms.txt
timeout 1500 python -m pytest tests/test_gpu_lfplus_scale.py tests/test_gpu_lfplus_protocol.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r5i_lfp_tests.txt
