"""(GPU box) stand-alone duration of the commit kernel at the BASELINE shape (kappa x n, batch): python tools/time_commit.py [ring] [kappa] [log2 n] [batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latticefold_amd import api
ring = sys.argv[1] if len(sys.argv) > 1 else "goldilocks"
kappa = int(sys.argv[2]) if len(sys.argv) > 2 else 26
lg = int(sys.argv[3]) if len(sys.argv) > 3 else 20
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 15
ctx = api.Context(0, ring=ring)
n = 1 << lg
sch = api.AjtaiCommitmentScheme(ctx, kappa=kappa, n=n, seed=7)
rng = np.random.default_rng(1)
P = 0xFFFFFFFF00000001 if ring == "goldilocks" else 15 * 2**27 + 1
f = rng.integers(0, P, size=(batch, n, ctx.RE), dtype=np.uint64)
for it in range(3):
    sch.commit_ntt(f)
    ks = ctx.kernel_stats()
    print(ring, "kappa", kappa, "n 2^%d" % lg, "batch", batch, "k_ajtai %.3f ms x %d" % (ks["ajtai_ms"], ks["ajtai_launches"]))
