cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_bb.py -x -q 2>&1 | tail -3) > gpurun_out/r04r.txt
for e in A=1 A=1; do env $e timeout 300 python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 $e ms/step %.3f'%d['ms_per_step'], d['phases_ms_per_step'])" >> gpurun_out/r04r.txt; done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_c3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
f=$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r04r_c3_kernel_stats.csv
cd $GRAFT_REPO_ROOT; python - <<'PY' >> gpurun_out/r04r.txt
import csv
rows=list(csv.DictReader(open('gpurun_out/r04r_c3_kernel_stats.csv')))
for r in rows[:30]:
    n=r['Name'].replace('void ','').replace('lfbb::','bb::').replace('lf::','')[:90]
    print("%7.3f ms/step  %5.1f launches/step  %s"%(float(r['TotalDurationNs'])/7/1e6, int(r['Calls'])/7, n))
PY
cat gpurun_out/r04r.txt
