"""BASELINE-size checks on the GPU through size-independent properties (no oracle run at these sizes):
  * the folded witness opens the folded commitment: A * f_0 == cm_0 (commitment homomorphism over the whole step),
  * decomposition proofs recompose: sum_k b^k y_k == cm, sum_k b^k v_k == v, sum_k b^k u_k == u,
  * every sumcheck round message satisfies the verifier recurrence p_i(0) + p_i(1) == p_{i-1}(r_{i-1}) (via the oracle's
    restated NIFSVerifier on the O(proof-size) data), and the folded witness norm stays below B/2."""
import numpy as np
import pytest

import lfo
from latticefold_amd import api
from latticefold_amd.workload import P, RE, make_workload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["C1", "T14", "C2"])
def test_fold_step_properties_at_scale(name):
    wl = make_workload(name)
    ctx = api.Context(0)
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        assert (wit.w_ccs == wl.w_ccs).all()
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, api.PoseidonTranscript())
        # (1) verifier recurrences on the proof (host-sized work only)
        inst = lfo.Instance(wl)
        rc, lc_v = inst.verify(lfo.Transcript(), acc, cccs, proof)
        assert rc == 0 and (lc_v == lc).all()
        # (2) commitment homomorphism: commit(f_0) == cm_0, computed by the GPU commit on the folded witness
        cm0 = lc[wl.s + 3: wl.s + 3 + wl.kappa]
        assert (w0.commit(scheme) == cm0).all()
        # (3) norm of the folded witness (so it can be folded again), via lf_linf_check on its NTT form
        ok, mx = ctx.linf_check(w0.f, wl.B // 2)
        assert ok, mx
        # (4) deterministic
        lc2, w02, proof2 = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, api.PoseidonTranscript())
        assert (proof2 == proof).all()
    finally:
        ctx.close()
