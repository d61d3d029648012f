#!/bin/bash
# (GPU box, ONE GPU) the N > 1 flow of bench.py with two gloo ranks sharing the GPU: (1) sharded headline + replicas extra, (2) a forced failure of rank 1's
# sharded measurement: both ranks must agree on the replicas fallback and leave with exit code 0
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/bench_fallback.txt; : > $out
run() {
  LF_FORCE_DEVICE=0 LF_DIST_BACKEND=gloo "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $PORT \
     bench.py --gpus 2 --steps 3 --warmup 1 --workload C2 --no-lfplus --no-cpu-baseline 2> gpurun_out/bench_fallback.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['scaling'], round(d['value'], 1), d['config'].get('parallelism', '')[:40], '| replicas:', (d.get('replicas') or {}).get('value'), '| note:', d.get('note'))"
  echo "exit code ${PIPESTATUS[0]}"
}
PORT=29541; echo "== normal" | tee -a $out; run env | tee -a $out
PORT=29542; echo "== rank 1 fails" | tee -a $out; run env LF_BENCH_FORCE_SHARD_FAIL=1 LF_SHARD_TIMEOUT=30 LF_SHARD_AGREE_TIMEOUT=60 | tee -a $out
tail -3 gpurun_out/bench_fallback.err | cut -c1-300 | tee -a $out
