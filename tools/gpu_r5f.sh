#!/bin/bash
# (GPU box) per-rank compute time of a G-way sharded C4 step with the model transport (tools/shard_model.py), and the step timeline of rank 0 of 8
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python tools/shard_model.py --workload C4 --worlds 1,2,4,8 > gpurun_out/r5f_shard_model_c4.txt 2> gpurun_out/r5f_shard_model_c4.err
grep "^#" gpurun_out/r5f_shard_model_c4.txt; tail -3 gpurun_out/r5f_shard_model_c4.err
LF_TIMELINE=1 timeout 300 python tools/shard_model.py --workload C4 --worlds 8 --steps 1 --warmup 3 2>&1 | grep "^\[timeline\]" | tail -45 > gpurun_out/r5f_timeline_c4_g8.txt
LF_TIMELINE=1 timeout 300 python tools/shard_model.py --workload C4 --worlds 2 --steps 1 --warmup 3 2>&1 | grep "^\[timeline\]" | tail -45 > gpurun_out/r5f_timeline_c4_g2.txt
timeout 600 python tools/shard_model.py --workload C4 --worlds 8 --ranks 3,7 | grep "^#"
