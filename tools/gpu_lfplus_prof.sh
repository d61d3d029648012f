#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_lfplus_prof.sh <tag> [nvars]
# PlusProver::prove wall numbers + rocprofv3 --kernel-trace --stats of the same command; results -> gpurun_out/
tag=$1; nv=${2:-18}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python $R/tools/bench_lfplus.py --cpu > $R/gpurun_out/lfplus_bench_${tag}.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lfp_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lfp_$tag -o p -- python $R/tools/bench_lfplus.py --nvars $nv --rounds 3 >/dev/null 2>&1
f=$(find /tmp/prof_lfp_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/lfplus_ks_${tag}_n${nv}.csv
cat $R/gpurun_out/lfplus_bench_${tag}.txt
head -25 $R/gpurun_out/lfplus_ks_${tag}_n${nv}.csv | cut -c1-150
