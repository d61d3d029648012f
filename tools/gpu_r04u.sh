cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_c3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r04u_c3_kernel_stats.csv
cd $GRAFT_REPO_ROOT; python - <<'PY' > gpurun_out/r04u.txt
import csv
rows=list(csv.DictReader(open('gpurun_out/r04u_c3_kernel_stats.csv')))
for r in rows[:60]:
    n=r['Name'].replace('void ','').replace('lfbb::','bb::').replace('lf::','')[:90]
    print("%7.3f ms/step  %5.1f launches/step  %s"%(float(r['TotalDurationNs'])/7/1e6, int(r['Calls'])/7, n))
PY
cat gpurun_out/r04u.txt
