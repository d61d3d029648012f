#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/s3z_i8x.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_ajtai_i8.py -q -m gpu -x -k "B6 or B10 or BDP or B21 or B32 or B14" 2>&1 | tail -4 | tee -a $out
timeout 900 python -m pytest tests/test_gpu_bb.py -q -m gpu -x -k "fold_step_parity or ajtai" 2>&1 | tail -3 | tee -a $out
timeout 900 python -m pytest tests/test_gpu_parity_scale.py -q -m gpu -x -k "C3 or B14" 2>&1 | tail -3 | tee -a $out
run() { local label="$1"; shift
  for rep in 1 2 3; do
    env "$@" python bench.py --workload C3 --steps 10 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label', round(d['ms_per_step'],3), 'commit ms', round(r['alg_bytes_per_launch']/r['achieved']/1e6,3) if r.get('achieved') else None, 'frac', round(r.get('frac') or 0,3), d['config'].get('matches_oracle_fixture'))" | tee -a $out
  done
}
run i8x X=1
run old LF_I8_NO_SPLIT=1
run i8x X=1
run old LF_I8_NO_SPLIT=1
python tools/i8_prof.py C3 2>&1 | tail -10 | tee -a $out
