#!/bin/bash
# (GPU box) LF+ parity tests + P20 resident prove, batched vs all-table sumcheckers of Cm::prove; tag = $1
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-cm}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_prover.py tests/test_gpu_lfplus_scale.py tests/test_dist_shard_lfplus.py -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
tail -5 gpurun_out/${tag}_tests.log
for v in ${2:-batched full}; do
  if [ $v = full ]; then export LFPLUS_CM_FULL=1; else unset LFPLUS_CM_FULL; fi
  timeout 600 python tools/bench_lfplus.py --nvars 20 --rounds 3 --k 4 --fresh 3 --resident > gpurun_out/${tag}_p20_$v.txt 2>&1
  grep -E "gpu_prove_ms" gpurun_out/${tag}_p20_$v.txt | cut -c1-260
  LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --rounds 1 --k 4 --fresh 3 --resident 2>&1 | grep -E "cm: " | tail -8
done
