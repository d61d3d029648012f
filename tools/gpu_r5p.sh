#!/bin/bash
# (GPU box) LatticeFold+ column-sharded prover: one rank's share of a G-way PlusProver::prove at 2^20 rows with the model transport
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python tools/shard_model.py --lfplus P20 --worlds 1,2,4,8 --steps 2 --warmup 1 > gpurun_out/r5p_shard_model_lfplus.txt 2> gpurun_out/r5p_err.txt
grep "^#" gpurun_out/r5p_shard_model_lfplus.txt; tail -3 gpurun_out/r5p_err.txt
