cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_c3
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
f=$(find /tmp/prof_c3 -name '*kernel_trace.csv' | head -1)
cd $GRAFT_REPO_ROOT; python - "$f" <<'PY' > gpurun_out/r04z.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find last k_fold_witness, walk back to the previous one
idx=[i for i,r in enumerate(rows) if 'k_fold_witness' in r['Kernel_Name']]
a,b=idx[-2]+1,idx[-1]+1
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n=r['Kernel_Name'].replace('void ','').replace('lfbb::','bb::').replace('lf::','')
    n=n.split('(')[0][:60]
    s=(int(r['Start_Timestamp'])-t0)/1e6; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print("%8.3f ms  %8.1f us  q%s  %s  grid %s"%(s,d,r.get('Queue_Id','?'),n,r.get('Grid_Size_X','?')))
PY
tail -5 gpurun_out/r04z.txt
