"""(GPU box) durations of the folding-sumcheck round kernels of the last step in a rocprofv3 kernel trace, in launch order:
python tools/round_durs.py kernel_trace.csv"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_fold_round" in r[2]]
last = idx[-24:]
for i in last:
    r = rows[i]
    print("%-34s %8.1f us  grid %s x %s x %s / %s" % (r[2].split("(")[0][-34:], (r[1] - r[0]) / 1e3, r[3], r[4], r[5], r[6]))
