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
names = ["[0] digits (wait d)", "[1] LDS-DMA issue", "[2] vectors", "[3] K-steps", "[4] -", "[5] wait for tile T+1", "[6] barrier"]
print("workload", wl.name if hasattr(wl, "name") else sys.argv[1:2], "witness" if wit_mode else "general", "tiles of workgroup 0:", a[:, 7], " kernel stats:", ctx.kernel_stats())
print("cycles per tile and wave (shader clock), waves 0..7 (0-3 multiply, 4-6 build vectors, 7 copies A):")
for i, n in enumerate(names):
    print("  %-24s" % n, " ".join("%7.0f" % (a[w, i] / max(a[w, 7], 1)) for w in range(8)))
a4 = a[:, 4].copy(); a[:, 4] = 0
print("  %-24s" % "sum", " ".join("%7.0f" % (a[w, :7].sum() / max(a[w, 7], 1)) for w in range(8)))
if a4[1] > 0:
    cyc = a[0, :7].sum()
    print("loop of workgroup 0: %.1f us = %.0f MHz shader clock; slowest workgroup %.1f us; mean %.1f us (100 MHz real-time counter)" % (a4[1] / 100, cyc / (a4[1] / 100), a4[2] / 100, a4[3] / 100 / 256))
wg = (C.c_uint32 * 512)()
lib.lfdbg_i8g_wg.argtypes = [C.POINTER(C.c_uint32)]
if lib.lfdbg_i8g_wg(wg) == 0:
    t = np.array(wg[:], dtype=np.float64) / 100
    n = int((t > 0).sum())
    t = t[:n]
    print("workgroups", n, "loop us: min %.0f median %.0f max %.0f" % (t.min(), np.median(t), t.max()))
    half = n // 2
    for name, sel in (("first half of the grid", t[:half]), ("second half", t[half:])):
        print("  %s: mean %.0f min %.0f max %.0f" % (name, sel.mean(), sel.min(), sel.max()))
    print("  by blockIdx %% 8 (XCD): " + " ".join("%.0f" % t[x::8].mean() for x in range(8)))
    print("  first 32:", " ".join("%.0f" % x for x in t[:32]))
    print("  sorted deciles:", " ".join("%.0f" % x for x in np.percentile(t, [0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 100])))
