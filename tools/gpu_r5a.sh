#!/bin/bash
# (GPU box) round-5 first call: the hygiene batch on hardware -- LF+ tests touched by it, the default bench line (fixture tie, LF+ warm timings), quick C2/C3 lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_prover.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5a_lfp_tests.txt
( time python bench.py ) > gpurun_out/r5a_bench_c4.json 2> gpurun_out/r5a_bench_c4.err
python bench.py --workload C2 --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus > gpurun_out/r5a_bench_c2.json 2>/dev/null
python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus > gpurun_out/r5a_bench_c3.json 2>/dev/null
LF_TIMELINE=1 python bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[timeline\]" | tail -40 > gpurun_out/r5a_timeline_c4.txt
cat gpurun_out/r5a_lfp_tests.txt; tail -3 gpurun_out/r5a_bench_c4.err
python - <<'PY'
import json
for w in ("c4","c2","c3"):
    try:
        d=json.loads(open(f"gpurun_out/r5a_bench_{w}.json").read().strip().splitlines()[-1])
        print(w, d["ms_per_step"], d["config"].get("matches_oracle_fixture"), d["roofline"]["frac"], (d.get("lfplus") or {}).get("runs") and [(r["workload"][:4], r["ms"], r["ms_host_io"], r.get("matches_oracle_fixture")) for r in d["lfplus"]["runs"]])
    except Exception as e: print(w, "failed", e)
PY
