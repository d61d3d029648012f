"""(GPU box) stand-alone duration of the GENERAL commit (AjtaiCommitmentScheme::commit_ntt, batch 1: the reference's "CommitNTT" bench, benches/ajtai.rs:15-31)
and of Witness::commit from a resident handle:  python tools/time_commit_general.py [workload] [kappa] [reps]
prints one JSON line: kernel ms (HIP events around the launches), SURVEY 8(d) bytes (kappa + 1) N E and the fraction of the 8 TB/s HBM peak"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latticefold_amd import api
from latticefold_amd.workload import make_workload
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
wl = make_workload(name)
kappa = int(sys.argv[2]) if len(sys.argv) > 2 else wl.kappa
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ring, n = wl.ring, wl.N
ctx = api.Context(0, ring=ring)
ctx.load_ccs(wl)
E = 192 if ring == "goldilocks" else 288
sch = api.AjtaiCommitmentScheme(ctx, kappa=kappa, n=n, seed=wl.ajtai_seed())
rng = np.random.default_rng(1)
P = 0xFFFFFFFF00000001 if ring == "goldilocks" else 15 * 2**27 + 1
f = rng.integers(0, P, size=(n, ctx.RE), dtype=np.uint64)
ms = []
for it in range(reps):
    sch.commit_ntt(f)
    ms.append(ctx.kernel_stats()["ajtai_ms"])
w = api.Witness.from_w_ccs(ctx, wl.w_ccs)
wms = []
for it in range(reps):
    t0 = time.perf_counter()
    w.commit(sch)
    wall = (time.perf_counter() - t0) * 1e3
    wms.append((ctx.kernel_stats()["ajtai_ms"], round(wall, 3)))
alg = (kappa + 1) * n * E
out = {"workload": name, "ring": ring, "kappa": kappa, "n": n, "commit_ntt_kernel_ms": ms, "alg_bytes": alg, "hbm_frac": alg / (min(ms) * 1e-3) / 8e12,
       "witness_commit_kernel_ms_wall_ms": wms, "witness_hbm_frac": alg / (min(x[0] for x in wms) * 1e-3) / 8e12}
print(json.dumps(out))
