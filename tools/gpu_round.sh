#!/bin/bash
# (GPU box) the judged artefacts of a round: default bench line (C4 with cpu_baseline), C2 / C3 lines, rocprofv3 kernel stats of the
# same command, HBM traffic from separate PMC passes.  usage: tools/gpu_round.sh <tag>   -> gpurun_out/<tag>_*
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r02}
mkdir -p $R/gpurun_out
python $R/bench.py > $R/gpurun_out/${tag}_bench_c4.json 2> $R/gpurun_out/${tag}_bench_c4.err
python $R/bench.py --workload C2 --steps 20 --warmup 3 > $R/gpurun_out/${tag}_bench_c2.json 2>/dev/null
python $R/bench.py --workload C3 --steps 10 --warmup 2 > $R/gpurun_out/${tag}_bench_c3.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for wl in C4 C3; do
  rm -rf /tmp/prof_$wl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
  f=$(find /tmp/prof_$wl -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/${tag}_$(echo $wl | tr A-Z a-z)_kernel_stats.csv
done
bash $R/tools/gpu_pmc.sh $tag C4
bash $R/tools/gpu_pmc.sh $tag C3
LF_TIMELINE=1 python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[timeline\]" | tail -32 > $R/gpurun_out/${tag}_timeline_c4.txt
LF_TIMELINE=1 python $R/bench.py --workload C3 --steps 1 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[bb timeline\]" | tail -36 > $R/gpurun_out/${tag}_timeline_c3.txt
bash $R/tools/gpu_sq.sh $tag C4
bash $R/tools/gpu_step_trace.sh $tag C3
bash $R/tools/gpu_step_trace.sh $tag C4
# LatticeFold+ at 2^20 rows: stage timeline and kernel stats
cd $R; LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 --resident 2>&1 | grep -v "round " | tail -26 > gpurun_out/${tag}_lfplus_p20.txt
cd /tmp; rm -rf /tmp/prof_lfp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lfp -o p -- python $R/tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 2 --resident >/dev/null 2>&1
f=$(find /tmp/prof_lfp -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/${tag}_lfplus_ks_p20.csv
python $R/tools/write_profile_manifest.py $tag $R/gpurun_out   # the manifest bench.py reads (copy it into profiles/ together with the <tag>_* files)
