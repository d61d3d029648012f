#!/usr/bin/env python3
"""(GPU box) per-phase shader-clock profile of the general commit kernel k_ajtai_i8g (LF_I8G_PROF instantiation): python tools/i8g_prof.py [workload] [witness]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LF_I8G_PROF"] = "1"
import numpy as np
from latticefold_amd import api
from latticefold_amd.workload import make_workload
wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "C4")
wit_mode = len(sys.argv) > 2 and sys.argv[2] == "witness"
ctx = api.Context(0, ring=wl.ring)
ctx.load_ccs(wl)
scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=7)
if wit_mode:
    w = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    for _ in range(3):
        w.commit(scheme)
else:
    P = 0xFFFFFFFF00000001
    f = np.random.default_rng(1).integers(0, P, size=(wl.N, ctx.RE), dtype=np.uint64)
    for _ in range(3):
        scheme.commit_ntt(f)
out = (C.c_uint64 * 64)()
lib = api._lib()
lib.lf_debug_i8_prof.argtypes = [C.POINTER(C.c_uint64)]
assert lib.lf_debug_i8_prof(out) == 0
a = np.array(out[:], dtype=np.float64).reshape(8, 8)
names = ["[0] digits (wait d)", "[1] load issue 1", "[2] vectors", "[3] K-steps", "[4] load issue 2", "[5] wait tile + LDS st", "[6] barrier"]
print("workload", wl.name if hasattr(wl, "name") else sys.argv[1:2], "witness" if wit_mode else "general", "tiles of workgroup 0:", a[:, 7], " kernel stats:", ctx.kernel_stats())
print("cycles per tile and wave (shader clock), waves 0..7 (0-3 multiply, 4-7 produce):")
for i, n in enumerate(names):
    print("  %-24s" % n, " ".join("%7.0f" % (a[w, i] / max(a[w, 7], 1)) for w in range(8)))
print("  %-24s" % "sum", " ".join("%7.0f" % (a[w, :7].sum() / max(a[w, 7], 1)) for w in range(8)))
