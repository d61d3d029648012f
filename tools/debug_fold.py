import sys, os, faulthandler
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
faulthandler.dump_traceback_later(280, exit=True)
import time
import numpy as np
from latticefold_amd import api
from latticefold_amd.workload import make_workload
name = sys.argv[1] if len(sys.argv) > 1 else "T8"
wl = make_workload(name)
ctx = api.Context(0); ctx.load_ccs(wl)
scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
cm = wit.commit(scheme); cccs = np.concatenate([cm, wl.x_ccs])
acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
print("linearized", flush=True)
for it in range(3):
    t0 = time.time()
    lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, api.PoseidonTranscript())
    dt = time.time() - t0
    print("folded", name, "wall %.1f ms" % (dt * 1e3), {k: round(v, 2) for k, v in ctx.phase_ms().items()}, ctx.kernel_stats(), flush=True)
    w0.free()
