#!/bin/bash
# (GPU box) kernel stats of the P20 resident LF+ prove + stage marks; tag = $1
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-ks}
mkdir -p $R/gpurun_out
cd $R; LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 --resident 2>&1 | grep -v "^\[lfplus\]   " | tail -30 > gpurun_out/${tag}_lfplus_p20.txt
LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 --resident 2>&1 | grep "cm round" | tail -40 > gpurun_out/${tag}_lfplus_cmrounds.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_lfp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lfp -o p -- python $R/tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 2 --resident >/dev/null 2>&1
f=$(find /tmp/prof_lfp -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/${tag}_lfplus_ks_p20.csv
cat $R/gpurun_out/${tag}_lfplus_p20.txt
head -30 $R/gpurun_out/${tag}_lfplus_ks_p20.csv | cut -c1-200
