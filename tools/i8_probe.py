#!/usr/bin/env python3
"""(GPU box) one decomposition (digit-plane commits) at a given workload, for a rocprofv3 kernel trace of k_ajtai_i8"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from latticefold_amd import api
from latticefold_amd.workload import make_workload
wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "C2")
ctx = api.Context(0)
ctx.load_ccs(wl)
scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=7)
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
for _ in range(3):
    api.LFDecompositionProver.prove(ctx, acc, wit, api.PoseidonTranscript())
