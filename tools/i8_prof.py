#!/usr/bin/env python3
"""(GPU box) per-phase shader-clock profile of k_ajtai_i8 (LF_I8_PROF instantiation): one fold step at C4, then the totals of workgroup 0"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LF_I8_PROF"] = "1"
import numpy as np
from latticefold_amd import api
from latticefold_amd.workload import make_workload
wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "C4")
ctx = api.Context(0, ring=wl.ring)
ctx.load_ccs(wl)
scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=7)
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript(ring=wl.ring))
for _ in range(2):
    api.NIFSProver.prove(ctx, acc, wit, cccs, wit, api.PoseidonTranscript(ring=wl.ring))
out = (C.c_uint64 * 64)()
lib = api._lib()
lib.lf_debug_i8_prof.argtypes = [C.POINTER(C.c_uint64)]
assert lib.lf_debug_i8_prof(out) == 0
a = np.array(out[:], dtype=np.float64).reshape(8, 8)
names = ["[0] handshake | DMA issue", "[1] digits", "[2] vectors", "[3] K-steps", "[4] (workgroup statistics)", "[5] wait for tile T+1", "[6] barrier"]
print("tiles per workgroup:", a[:, 7], " kernel stats:", ctx.kernel_stats())
print("cycles per tile and wave (shader clock), waves 0..7 (0-3 multiply, 4-6 digits + vectors, 7 copies A; i8x: 4-7 produce):")
for i, n in enumerate(names):
    print("  %-22s" % n, " ".join("%7.0f" % (a[w, i] / max(a[w, 7], 1)) for w in range(8)))
print("  %-22s" % "sum", " ".join("%7.0f" % ((a[w, :4].sum() + a[w, 5:7].sum()) / max(a[w, 7], 1)) for w in range(8)))
if a[5, 4] > 0:    # k_ajtai_i8s: per-workgroup loop statistics of the LAST profiled launch (column 4)
    nwg, tiles = a[5, 4], max(a[0, 7], 1)
    cmax, cmin, cmean = a[1, 4], float(2**62) - a[2, 4], a[3, 4] / nwg
    print("workgroups %d: loop cycles per tile min / mean / max = %.0f / %.0f / %.0f;  workgroup 0: %.1f us at the 100 MHz counter -> shader clock %.3f GHz;  slowest workgroup %.1f us"
          % (nwg, cmin / tiles, cmean / tiles, cmax / tiles, a[0, 4] / 100.0, (a[0, :4].sum() + a[0, 5:7].sum()) / max(a[0, 4], 1) / 10.0, a[4, 4] / 100.0))
