#!/bin/bash
# (GPU box) judged artefacts of the LatticeFold+ double-commitment row: bench lines (with the CPU oracle beside them) and the rocprofv3
# kernel stats of the same command.   usage: tools/gpu_dc.sh <tag>   -> gpurun_out/<tag>_dc_*
tag=${1:-r02}
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
python $R/tools/bench_double_commitment.py > $R/gpurun_out/${tag}_dc_bench.jsonl 2>/tmp/dc_err.txt || tail -5 /tmp/dc_err.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dc -o p -- python $R/tools/bench_double_commitment.py --no-cpu --only 131072 >/dev/null 2>&1
f=$(find /tmp/prof_dc -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_dc_kernel_stats_n131072.csv
cat $R/gpurun_out/${tag}_dc_bench.jsonl | cut -c1-330
head -8 $R/gpurun_out/${tag}_dc_kernel_stats_n131072.csv
