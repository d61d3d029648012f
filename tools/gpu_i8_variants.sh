#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/i8_variants.txt; : > $out
run() { local label="$1"; shift
  for rep in 1 2; do
    env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],3), 'commit ms', round(d['roofline']['alg_bytes_per_launch']/d['roofline']['achieved']/1e6,3), 'frac', round(d['roofline']['frac'],3), d['config'].get('matches_oracle_fixture'))" | tee -a $out
  done
  env "$@" python tools/i8_prof.py C4 2>&1 | tail -10 | grep -E "ajtai_ms|K-steps|sum" | cut -c1-200 | tee -a $out
}
run default X=1
run nobits LF_I8_BITS=0
run nobits_nocols LF_I8_BITS=0 LF_I8_COLS=0
run nocols LF_I8_COLS=0
run default_again X=1
