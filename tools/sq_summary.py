#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ counters (counter_collection CSV) per kernel: totals over all launches and the
fractions of wave time spent issuing VALU / any instruction / waiting.  SQ_WAVE_CYCLES, SQ_ACTIVE_INST_*, SQ_WAIT_* count
quad-cycles summed over waves (MI355X_MICROARCH.md); the fractions are ratios of like units."""
import csv, json, re, sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = re.sub(r"^void\s+", "", r["Kernel_Name"])
        k = re.sub(r"\(.*$", "", k).replace("lf::", "").replace("lfbb::", "bb::")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r.get("Dispatch_Id"))
out = {"method": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                 f"SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace (bench.py --workload {sys.argv[2]} --steps 1 --warmup 1); sums over the "
                 "launches of each kernel; fractions are of SQ_WAVE_CYCLES",
       "kernels": {}}
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = c.get("SQ_WAVE_CYCLES", 0)
    if w <= 0:
        continue
    out["kernels"][k] = {"launches": len(launches[k]), "wave_cycles": w,
                         "valu_active_frac": c.get("SQ_ACTIVE_INST_VALU", 0) / w, "any_active_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / w,
                         "lds_active_frac": c.get("SQ_ACTIVE_INST_LDS", 0) / w, "wait_any_frac": c.get("SQ_WAIT_ANY", 0) / w,
                         "wait_inst_any_frac": c.get("SQ_WAIT_INST_ANY", 0) / w, "valu_insts": c.get("SQ_INSTS_VALU", 0),
                         "busy_cycles": c.get("SQ_BUSY_CYCLES", 0)}
print(json.dumps(out, indent=1))
