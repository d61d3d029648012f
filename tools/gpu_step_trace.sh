#!/bin/bash
# (GPU box) every kernel launch of ONE fold step in start order: start (ms since the step's first kernel), duration, queue, name, grid -- from a rocprofv3 kernel trace
# of bench.py.  usage: tools/gpu_step_trace.sh <tag> [workload]   -> gpurun_out/<tag>_step_trace_<wl>.txt   (times are stretched by the tracer: use them for the
# order and the relative sizes; wall-clock marks come from LF_TIMELINE=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-rXX}; wl=${2:-C3}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tr -o p -- python $R/bench.py --workload $wl --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus --no-ajtai --no-shard-model --chain 0 >/dev/null 2>&1
f=$(find /tmp/prof_tr -name '*kernel_trace.csv' | head -1)
cd $R; python - "$f" <<'PY' > gpurun_out/${tag}_step_trace_$(echo $wl | tr A-Z a-z).txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_fold_witness' in r['Kernel_Name']]   # the last kernel of a step
a,b=idx[-2]+1,idx[-1]+1
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n=r['Kernel_Name'].replace('void ','').replace('lfbb::','bb::').replace('lf::','').split('(')[0][:60]
    s=(int(r['Start_Timestamp'])-t0)/1e6; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print("%8.3f ms  %8.1f us  q%s  %s  grid %s"%(s,d,r.get('Queue_Id','?'),n,r.get('Grid_Size_X','?')))
PY
