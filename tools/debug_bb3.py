import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
variant = sys.argv[1]
if "G" in variant:
    import lfo   # Goldilocks oracle module imported first (as test_gpu_scale.py does)
import lfo_bb
from latticefold_amd import api
from latticefold_amd.workload import make_workload
wl = make_workload("B10")
ctx = api.Context(0, ring="babybear")
ctx.load_ccs(wl)
scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
if "W" in variant:
    assert (wit.w_ccs == wl.w_ccs).all()
cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
tr = lambda: api.PoseidonTranscript(ring="babybear")
acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
inst = lfo_bb.Instance(wl)
rc, lc_v = inst.verify(lfo_bb.Transcript(), acc, cccs, proof)
print(variant, "verify rc", rc)
