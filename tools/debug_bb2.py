import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lfo_bb as lfo
from latticefold_amd import api
from latticefold_amd.workload import make_workload
name = sys.argv[1]
gen = sys.argv[2] == "gen"
wl = make_workload(name)
ctx = api.Context(0, ring="babybear")
ctx.load_ccs(wl)
inst = lfo.Instance(wl)
if gen:
    scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
else:
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
cm = wit.commit(scheme)
A = wl.ajtai_matrix()
f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
print("commit eq oracle", (cm == lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_coeff))).all())
cccs = np.concatenate([cm, wl.x_ccs])
tr = lambda: api.PoseidonTranscript(ring="babybear")
acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
for rep in range(3):
    lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
    rc, lc_v = inst.verify(lfo.Transcript(), acc, cccs, proof)
    print("rep", rep, "verify rc", rc)
