#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/s4c_ab.txt; : > $out
run() { local label="$1"; shift
    env "$@" python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],3), 'commit ms', round(d['roofline']['alg_bytes_per_launch']/d['roofline']['achieved']/1e6,3))" | tee -a $out
}
for rep in 1 2 3 4 5; do
  run default X=1
  run cols LF_I8_COLS=1
  run bits_cols LF_I8_BITS=1 LF_I8_COLS=1
done
