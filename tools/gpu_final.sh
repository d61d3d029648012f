#!/bin/bash
# (GPU box) refresh the judged artefacts: default bench line (C4, with cpu_baseline), C2 and C3 lines, kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r01_d}
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/${tag}_bench_c4.json 2> $R/gpurun_out/${tag}_bench_c4.err
python $R/bench.py --workload C2 --steps 20 --warmup 3 > $R/gpurun_out/${tag}_bench_c2.json 2>/dev/null
python $R/bench.py --workload C3 --steps 5 --warmup 2 > $R/gpurun_out/${tag}_bench_c3.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for wl in C4 C3; do
  rm -rf /tmp/prof_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
  f=$(find /tmp/prof_$wl -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/${tag}_$(echo $wl | tr A-Z a-z)_kernel_stats.csv
done
