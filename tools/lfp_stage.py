import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from math import ceil, log
from latticefold_amd import plus
P, D = plus.P, 16
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n, kappa, k, B = 1 << nv, 2, 2, 6186
ell = ceil(log(P) / log(8))
rng = np.random.default_rng(1)
A = rng.integers(0, P, size=(kappa, n, D), dtype=np.uint64)
r1cs = plus.r1cs_decomposed_square((plus.identity_csr(n // k),) * 3, n, B, k)
params = plus.PlusParameters(plus.LinParameters(kappa, plus.DecompParameters(8, k, ell)), B)
zs = []
for _ in range(2):
    z = np.zeros((n // k, D), dtype=np.uint64); z[:, 0] = rng.integers(0, 2, size=n // k); zs.append(z)
for rep in range(3):
    prover = plus.PlusProver.init(A, list(r1cs), 2, params, plus.PoseidonTranscript())
    comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, z, 1, B, k) for z in zs]
    T = {}
    t0 = time.perf_counter()
    ctxs = prover.ctxs[:2]
    lproof = []
    for i, ci in enumerate(comps):
        t1 = time.perf_counter(); ctxs[i].set_witness(ci.f); T["set_witness"] = T.get("set_witness", 0) + time.perf_counter() - t1
        t1 = time.perf_counter()
        keep, rp, cp, vp = plus._csr_args(plus.RESIDENT(3))
        msgs, ro, ev = np.zeros((nv, 4, D), dtype=np.uint64), np.zeros(nv, dtype=np.uint64), np.zeros((4, D), dtype=np.uint64)
        ctxs[i]._chk(plus._lib().lfplus_r1cs_linearize(ctxs[i].h, prover.transcript.h, rp, cp, vp, *[x.ctypes.data_as(plus.u64p) for x in (msgs, ro, ev)]))
        T["linearize"] = T.get("linearize", 0) + time.perf_counter() - t1
    t1 = time.perf_counter(); linb2x, cm = plus.mlin(ctxs, prover.transcript, params.lin, prover.res); T["mlin"] = time.perf_counter() - t1
    t1 = time.perf_counter(); dec = ctxs[0].decompose(None, None, B, plus._ro_pairs(linb2x["ro"]), prover.res); T["decompose"] = time.perf_counter() - t1
    T["total"] = time.perf_counter() - t0
    prover.close()
    print({k: round(v * 1e3, 2) for k, v in T.items()})
