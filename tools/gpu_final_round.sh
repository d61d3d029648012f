#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_round.sh r03d > gpurun_out/r03d_round.log 2>&1
bash tools/gpu_lfplus_prof.sh r03d 18 > gpurun_out/r03d_lfplus.log 2>&1
python - <<'PY' > gpurun_out/r03d_rebuild.txt 2>&1
import time, numpy as np
from latticefold_amd import api
from latticefold_amd.workload import make_workload
wl = make_workload("C4")
ctx = api.Context(0, ring=wl.ring); ctx.load_ccs(wl)
t0=time.perf_counter(); scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed(), digits_only=True); ctx.synchronize(); t1=time.perf_counter()
print("install digits-only: %.1f ms" % ((t1-t0)*1e3), "free/total GiB", [round(x/2**30,2) for x in ctx.device_memory()])
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
for i in range(3):
    t0=time.perf_counter(); cm = wit.commit(scheme); t1=time.perf_counter(); print("witness commit (NTT form rebuilt from the bytes): %.1f ms" % ((t1-t0)*1e3))
ctx.close()
ctx = api.Context(0, ring=wl.ring); ctx.load_ccs(wl)
t0=time.perf_counter(); scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed()); ctx.synchronize(); t1=time.perf_counter()
print("install both forms: %.1f ms" % ((t1-t0)*1e3), "free/total GiB", [round(x/2**30,2) for x in ctx.device_memory()])
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
for i in range(2):
    t0=time.perf_counter(); cm2 = wit.commit(scheme); t1=time.perf_counter(); print("witness commit (resident NTT form): %.1f ms" % ((t1-t0)*1e3))
print("same commitment:", bool((cm == cm2).all()))
PY
ls gpurun_out | grep r03d
# the bench lines once more, now that the profiles of this tag exist (bench.py reads profiles/<tag>_*): copy gpurun_out/<tag>_* to profiles/ first when run by hand
