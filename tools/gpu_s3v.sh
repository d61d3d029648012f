#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/s3v_cols.txt; : > $out
LF_I8_COLS=1 timeout 600 python -m pytest tests/test_gpu_ajtai_i8.py -q -m gpu -x 2>&1 | tail -2 | tee -a $out
LF_I8_COLS=1 timeout 600 python -m pytest tests/test_gpu_parity_scale.py -q -m gpu -x -k "C4 or c4" 2>&1 | tail -2 | tee -a $out
for rep in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export LF_I8_COLS=1; else unset LF_I8_COLS; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cols=$v', round(d['ms_per_step'],3), 'commit ms', round(d['roofline']['alg_bytes_per_launch']/d['roofline']['achieved']/1e6,3), 'frac', round(d['roofline']['frac'],3))" | tee -a $out
  done
done
unset LF_I8_COLS
LF_I8_COLS=1 python tools/i8_prof.py C4 2>&1 | tail -10 | tee -a $out
