#!/bin/bash
# (GPU box) hand-over thresholds of the sharded sumchecks against the per-rank model (tools/shard_model.py): G = 8 and 4, C4
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for G in 8 4; do
for lin in 4096 16384 65536 262144; do
for fold in 2048 8192 32768; do
echo "G=$G LIN_MIN=$lin FOLD_MIN=$fold: $(LF_SHARD_LIN_MIN=$lin LF_SHARD_FOLD_MIN=$fold timeout 300 python tools/shard_model.py --workload C4 --worlds $G --steps 5 2>/dev/null | grep '^#')"
done; done; done 2>&1 | tee gpurun_out/r5g_thresholds.txt
