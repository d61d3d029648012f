#!/bin/bash
# (GPU box) instruction-cache behaviour of the sumcheck round kernels: SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_IFETCH, SQC_ICACHE_REQ / _MISSES
# per kernel dispatch (one rocprofv3 --pmc pass, --kernel-trace only).  usage: tools/gpu_icache.sh [workload]
wl=${1:-C4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ic_pass
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/ic_pass -o p -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-lfplus >/dev/null 2>/tmp/ic_err.txt
f=$(find /tmp/ic_pass -name '*counter_collection.csv' | head -1)
if [ -z "$f" ]; then tail -5 /tmp/ic_err.txt; rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch" | head; exit 1; fi
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.OrderedDict()
for r in rows:
    key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-40:])
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
items = [(k, v) for k, v in by.items() if "k_fold_round<" in k[1]]
for k, v in items[-12:]:
    print(k[1], {n: int(x) for n, x in v.items()})
PY
