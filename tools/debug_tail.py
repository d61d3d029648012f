"""(GPU box) compare the fold-step proof with the persistent tail kernel against per-round launches, message by message"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latticefold_amd import api
from latticefold_amd.workload import make_workload

name = sys.argv[1] if len(sys.argv) > 1 else "T8"
wl = make_workload(name)
ctx = api.Context(0, ring=wl.ring)
ctx.load_ccs(wl)
scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
tr = lambda: api.PoseidonTranscript(ring=wl.ring)
acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
os.environ["LF_NO_TAIL"] = "1"
lc0, w0, p0 = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
del os.environ["LF_NO_TAIL"]
for rep in range(3):
    lc1, w1, p1 = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
    tau = wl.tau
    lin = wl.s * (wl.d + 2) + tau + wl.t
    dec = wl.K * (wl.t + tau + wl.l + 1 + wl.kappa)
    o = lin + 2 * dec
    bad = np.nonzero((p0 != p1).any(axis=1))[0]
    print("rep", rep, "differing proof elements:", bad[:10] - o, "of fold part (5 per round)")
    if bad.size:
        e = bad[0]
        print(" first diff element", e - o, "round", (e - o) // 5 + 1, "point", (e - o) % 5)
        print("  want", p0[e][:6]); print("  got ", p1[e][:6])
        print("  slots differing:", np.nonzero((p0[e] != p1[e]).reshape(8, 3).any(axis=1))[0])
os.environ["LF_THETA_EVAL"] = "1"
lc2, w2, p2 = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
print("with LF_THETA_EVAL: proof equal", (p2 == p0).all(), "lc equal", (lc2 == lc0).all(), "w equal", (w2.f == w0.f).all())
