#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/s3t_sweep.txt; : > $out
run() { local label="$1"; shift
  for rep in 1 2; do
    env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['phases_ms_per_step'].items() if k.startswith('fold')})" | tee -a $out
  done
}
run default X=1
run ct64k LF_FOLD_CHUNK_THREADS=65536
run ct256k LF_FOLD_CHUNK_THREADS=262144
run ct512k LF_FOLD_CHUNK_THREADS=524288
run ct1m LF_FOLD_CHUNK_THREADS=1048576
run ct2m LF_FOLD_CHUNK_THREADS=2097152
run ct4m LF_FOLD_CHUNK_THREADS=4194304
run tail4k LF_TAIL_N=4096
run tail1k LF_TAIL_N=1024
run default_again X=1
