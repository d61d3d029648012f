#!/bin/bash
# (GPU box) kernel times of the folding rounds with and without the split form of the table rounds: tools/gpu_ks_split.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in split nosplit; do
  rm -rf /tmp/p_$v
  if [ $v = nosplit ]; then export LF_FOLD_ROUNDS_NO_SPLIT=1; else unset LF_FOLD_ROUNDS_NO_SPLIT; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
  f=$(find /tmp/p_$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_fold_round<", "k_reduce_rows", "k_eq_pairsum", "k_fold_r4tab", "k_fold_r5tab", "k_fix<")):
        print(n[:60].ljust(60), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "total_ms/step", round(float(r["TotalDurationNs"]) / 7e6, 3))
PY
done
