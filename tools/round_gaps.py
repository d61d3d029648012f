"""(GPU box) per-round anatomy of the small folding-sumcheck rounds from a rocprofv3 kernel trace: period between consecutive
k_fold_round launches, GPU-busy time inside it, and the idle remainder (host transcript + launch/sync latency)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_fold_round<" in r[2]]
out = []
for a, b in zip(idx, idx[1:]):
    period = rows[b][0] - rows[a][0]
    busy = sum(r[1] - r[0] for r in rows[a:b])
    if period < 200000:      # small rounds only (< 200 us)
        out.append((period, busy, b - a))
out = out[-40:]
print("small rounds: n=%d  period %.1f us  gpu-busy %.1f us  kernels/round %.1f" % (len(out), sum(o[0] for o in out) / len(out) / 1e3,
      sum(o[1] for o in out) / len(out) / 1e3, sum(o[2] for o in out) / len(out)))
a, b = idx[-3], idx[-2]
for r in rows[a:b]:
    print("  %-40s start +%.1f us  dur %.1f us" % (r[2][:40], (r[0] - rows[a][0]) / 1e3, (r[1] - r[0]) / 1e3))
