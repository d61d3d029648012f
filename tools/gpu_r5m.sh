#!/bin/bash
# (GPU box) LatticeFold+ host-I/O form: device-side canonical check, uploads on a worker thread; tests, then the default bench line (lfplus extra: ms / ms_host_io)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_lfplus_prover.py tests/test_gpu_lfplus.py -q -m gpu -x 2>&1 | tail -6 | tee gpurun_out/r5m_lfp_tests.txt
python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r5m_bench_c4.json 2> gpurun_out/r5m_bench_c4.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5m_bench_c4.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], [(r["workload"][:3], round(r["ms"], 2), round(r["ms_host_io"], 2), r.get("matches_oracle_fixture")) for r in d["lfplus"]["runs"]])
PY
LFPLUS_SERIAL_UPLOADS=1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial uploads:', [(r['workload'][:3], round(r['ms'],2), round(r['ms_host_io'],2)) for r in d['lfplus']['runs']])"
