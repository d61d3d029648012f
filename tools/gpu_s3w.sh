#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/s3w_bits.txt; : > $out
run() { local label="$1"; shift
  for rep in 1 2; do
    env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],3), 'commit ms', round(d['roofline']['alg_bytes_per_launch']/d['roofline']['achieved']/1e6,3), 'frac', round(d['roofline']['frac'],3), d['config'].get('matches_oracle_fixture'))" | tee -a $out
  done
}
run default X=1
run cols LF_I8_COLS=1
run bits LF_I8_BITS=1
run bits_cols LF_I8_BITS=1 LF_I8_COLS=1
run bits_cols_w256 LF_I8_BITS=1 LF_I8_COLS=1 LF_I8_WGS=256
run default_again X=1
LF_I8_BITS=1 LF_I8_COLS=1 timeout 600 python -m pytest tests/test_gpu_ajtai_i8.py -q -m gpu -x 2>&1 | tail -2 | tee -a $out
LF_I8_BITS=1 LF_I8_COLS=1 timeout 600 python -m pytest tests/test_gpu_parity_scale.py -q -m gpu -x -k "C4 or c4" 2>&1 | tail -2 | tee -a $out
