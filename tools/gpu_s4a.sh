#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/s4a_c3_wgs.txt; : > $out
run() { local label="$1"; shift
  for rep in 1 2; do
    env "$@" python bench.py --workload C3 --steps 10 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label', round(d['ms_per_step'],3), 'commit ms', round(r['alg_bytes_per_launch']/r['achieved']/1e6,3), d['config'].get('matches_oracle_fixture'))" | tee -a $out
  done
}
run w224 X=1
run w192 LF_I8_WGS=192
run w208 LF_I8_WGS=208
run w240 LF_I8_WGS=240
run w256 LF_I8_WGS=256
run evals_first LF_BB_EVALS_FIRST=1
run w224 X=1
