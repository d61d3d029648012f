#!/bin/bash
# (GPU box) schedule sweep of the C4 step after the host-Poseidon speed-up (phase 1 is now GPU-bound): lane-1 order, z_R position, commit workgroups
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; out=gpurun_out/schedule_sweep.txt; : > $out
run() { # label, env...
  local label="$1"; shift
  for rep in 1 2; do
    env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],3))" | tee -a $out
  done
}
run default X=1
run commits_first LF_COMMITS_FIRST=1
run zr0 LF_ZR_POS=0
run zr1 LF_ZR_POS=1
run zr3 LF_ZR_POS=3
run wgs208 LF_I8_WGS=208
run wgs240 LF_I8_WGS=240
run wgs256 LF_I8_WGS=256
run two_stage LF_EVALS_TWO_STAGES=1
run no_early_y LF_NO_EARLY_Y=1
run cf_zr0 LF_COMMITS_FIRST=1 LF_ZR_POS=0
run cf_zr3 LF_COMMITS_FIRST=1 LF_ZR_POS=3
run cf_wgs240 LF_COMMITS_FIRST=1 LF_I8_WGS=240
run vs_back4 LF_LIN_VS_BACK=4
run vs_back8 LF_LIN_VS_BACK=8
run default_again X=1
