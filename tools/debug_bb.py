"""debug helper (GPU box): BabyBear linearization at a given size, first differing element vs the oracle"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lfo_bb as lfo
from latticefold_amd import api
from latticefold_amd.workload import make_workload

name = sys.argv[1] if len(sys.argv) > 1 else "B10"
wl = make_workload(name)
ctx = api.Context(0, ring="babybear")
ctx.load_ccs(wl)
inst = lfo.Instance(wl)
A = wl.ajtai_matrix()
scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
print("f_coeff eq", (wit.f_coeff == f_coeff).all())
cm = wit.commit(scheme)
print("commit eq", (cm == lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_coeff))).all())
cccs = np.concatenate([cm, wl.x_ccs])
z = wl.z()
for j in range(wl.t):
    got = ctx.mat_vec_mul(j, z)
    rows = min(wl.n, wl.m)
    prod = np.zeros((rows, 72), dtype=np.uint64)
    lfo.lib().lfo_ring_mul_ntt(lfo._p64(np.ascontiguousarray(wl.val[j])), lfo._p64(np.ascontiguousarray(z[:rows])), lfo._p64(prod), rows)
    print("spmv", j, (got[:rows] == prod).all(), not got[rows:].any())
acc_g, pr_g = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript(ring="babybear"))
acc_o, pr_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
bad = np.nonzero((pr_g != pr_o).any(axis=1))[0]
print("lin proof first bad elements", bad[:8], "of", pr_g.shape[0], "(4 per round)")
if bad.size:
    b = bad[0]
    print("words differing in elem", b, np.nonzero(pr_g[b] != pr_o[b])[0][:20])
# eq table check at this size
pt = np.array([[ (i*7+j*3+1) % lfo.P for j in range(9)] for i in range(wl.s)], dtype=np.uint64)
eq = ctx.build_eq(pt)
oeq = lfo.build_eq(np.tile(pt, (1, 8)))
print("eq", (eq == oeq[:, :9]).all())
tabs = np.stack([ctx.mat_vec_mul(j, z) for j in range(wl.t)])
got = ctx.evaluate_mles(tabs, pt)
print("mle", [(got[a] == lfo.mle_eval(tabs[a], np.tile(pt, (1, 8)))).all() for a in range(wl.t)])
if len(sys.argv) > 2:
    tr = lambda: api.PoseidonTranscript(ring="babybear")
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr())
    rc, lc_v = inst.verify(lfo.Transcript(), acc_g, cccs, proof_g)
    print("verify rc", rc)
    import time; t0 = time.time()
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    print("oracle fold s", time.time() - t0)
    lin = wl.s * (wl.d + 2) + 9 + wl.t
    dec = wl.K * (wl.t + 9 + wl.l + 1 + wl.kappa)
    bad = np.nonzero((proof_g != proof_o).any(axis=1))[0]
    print("sections: lin<%d decL<%d decR<%d fold" % (lin, lin + dec, lin + 2 * dec), "bad:", bad[:12], "count", bad.size)
    print("decL layout: u_s[K][t]=%d v_s[K][9]=%d x_s=%d y_s=%d" % (wl.K * wl.t, wl.K * 9, wl.K * (wl.l + 1), wl.K * wl.kappa))
    print("lc eq", (lc_g == lc_o).all(), "f0 eq", (w0.f == f0_o).all())
