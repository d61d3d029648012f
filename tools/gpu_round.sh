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
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -o p -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus --no-ajtai --no-shard-model --chain 0 >/dev/null 2>&1
  f=$(find /tmp/prof_$wl -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/${tag}_$(echo $wl | tr A-Z a-z)_kernel_stats.csv
done
bash $R/tools/gpu_pmc.sh $tag C4
# the general commit on its own (the reference's CommitNTT shape at C4's kappa, and Witness::commit): kernel stats + HBM traffic of k_ajtai_i8g
cd /tmp; rm -rf /tmp/prof_g
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_g -o p -- python $R/tools/time_commit_general.py C4 >/dev/null 2>&1
f=$(find /tmp/prof_g -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/${tag}_commit_general_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_g
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_g -o p -- python $R/tools/time_commit_general.py C4 26 2 >/dev/null 2>&1
  f=$(find /tmp/pmc_g -name '*counter_collection.csv' | head -1)
  python - "$f" $ctr >> $R/gpurun_out/${tag}_commit_general_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if r.get("Counter_Name") == sys.argv[2]:
        acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]:
    scale = 2.0 if sys.argv[2] == "FETCH_SIZE" else 1.0     # gfx950: FETCH_SIZE counts 64-byte requests as 32 (MI355X_MICROARCH.md); values are KiB
    print("%s %-60s launches %3d  mean %.1f MB per launch (raw KiB %.0f%s)" % (sys.argv[2], k, len(v), sum(v) / len(v) * 1024 * scale / 1e6, sum(v) / len(v), ", doubled" if scale == 2 else ""))
PY
done
cd /tmp
bash $R/tools/gpu_pmc.sh $tag C3
LF_TIMELINE=1 python $R/bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-lfplus --no-ajtai --no-shard-model --chain 0 2>&1 | grep "^\[timeline\]" | tail -32 > $R/gpurun_out/${tag}_timeline_c4.txt
LF_TIMELINE=1 python $R/bench.py --workload C3 --steps 1 --warmup 2 --no-cpu-baseline --no-lfplus --no-ajtai --no-shard-model --chain 0 2>&1 | grep "^\[bb timeline\]" | tail -36 > $R/gpurun_out/${tag}_timeline_c3.txt
bash $R/tools/gpu_sq.sh $tag C4
bash $R/tools/gpu_step_trace.sh $tag C3
bash $R/tools/gpu_step_trace.sh $tag C4
# LatticeFold+ at 2^20 rows: stage timeline and kernel stats
cd $R; LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 --resident 2>&1 | grep -v "round " | tail -26 > gpurun_out/${tag}_lfplus_p20.txt
cd /tmp; rm -rf /tmp/prof_lfp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lfp -o p -- python $R/tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 2 --resident >/dev/null 2>&1
f=$(find /tmp/prof_lfp -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/${tag}_lfplus_ks_p20.csv
python $R/tools/write_profile_manifest.py $tag $R/gpurun_out   # the manifest bench.py reads (copy it into profiles/ together with the <tag>_* files)
