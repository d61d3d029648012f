#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <tag> [bench args]
# runs bench.py twice (wall numbers) and once under rocprofv3 --kernel-trace --stats; results -> gpurun_out/
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
for i in 1 2; do
  python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-lfplus "$@" 2>/dev/null > $R/gpurun_out/bench_${tag}_$i.json
  python -c "
import json,sys;d=json.load(open('$R/gpurun_out/bench_${tag}_$i.json'));print('bench',d['value'],d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-lfplus "$@" >/dev/null 2>&1
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/ks_${tag}.csv
