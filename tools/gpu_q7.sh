cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bb.py -x -q 2>&1 | tail -3) > gpurun_out/q7.txt
for wl in C4 C4 C3 C3; do timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl ms/step %.3f'%d['ms_per_step'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})" >> gpurun_out/q7.txt; done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_q6
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q6 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload C4 --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
f=$(find /tmp/prof_q6 -name '*kernel_stats.csv' | head -1)
python - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/q7.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'spmv' in r['Name']: print(r['Name'][:40], r['Calls'], round(float(r['TotalDurationNs'])/int(r['Calls'])/1e3,1),'us avg  min',r['MinNs'],'max',r['MaxNs'])
PY
cat $GRAFT_REPO_ROOT/gpurun_out/q7.txt
