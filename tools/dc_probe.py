#!/usr/bin/env python3
"""(GPU box) kernel-time probe of the double commitment: from_f at several k and a plain commit, for a rocprofv3 kernel trace"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from latticefold_amd import plus
import lfp
n, kappa = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = plus.PlusContext(0)
A = lfp.splitmix(1, 0, kappa * n * 16).reshape(kappa, n, 16)
v = (lfp.splitmix(2, 0, n * 16) % np.uint64(15)).astype(np.int64) - 7
f = np.where(v < 0, np.uint64(plus.P) - (-v).astype(np.uint64), v.astype(np.uint64)).reshape(n, 16)
ctx.set_matrix(A); ctx.set_witness(f)
for k in (4, 2, 1):
    print("k", k, ctx.time_rg_from_f(plus.DecompParameters.for_frog(k), 3))
ctx.commit(f)
