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


@pytest.mark.parametrize("name", ["C1", "T14", "C2", "C4"])
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


def test_repeated_steps_are_stable_and_chain():
    """20 chained fold steps at C1 (IVC style: the folded accumulator/witness feed the next step): every step verifies,
    the witness norm stays below B/2, device memory does not grow (handles freed), results are reproducible."""
    wl = make_workload("C1")
    ctx = api.Context(0)
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        inst = lfo.Instance(wl)

        def chain(steps):
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
            w = wit
            digests = []
            for i in range(steps):
                lc, w_next, proof = api.NIFSProver.prove(ctx, acc, w, cccs, wit, api.PoseidonTranscript())
                rc, lc_v = inst.verify(lfo.Transcript(), acc, cccs, proof)
                assert rc == 0 and (lc_v == lc).all(), i
                ok, mx = ctx.linf_check(w_next.f, wl.B // 2)
                assert ok, (i, mx)
                digests.append(int(proof[-1, 0]) ^ int(lc[wl.s, 0]))
                if w is not wit:
                    w.free()
                acc, w = lc, w_next
            return digests

        d1 = chain(20)
        free1 = ctx.mem_info()[0]
        d2 = chain(20)
        free2 = ctx.mem_info()[0]
        assert d1 == d2
        assert abs(free1 - free2) < 64 << 20      # no growth beyond allocator noise
    finally:
        ctx.close()


@pytest.mark.parametrize("name", ["B10", "B14", "C3"])
def test_babybear_fold_step_properties_at_scale(name):
    """BASELINE configs[2] (BabyBearRingNTT, 2^18 rows, kappa = 16) and two smaller BabyBear cases: same size-independent
    properties as above, checked with the BabyBear build of the oracle's verifier on the O(proof-size) data."""
    import lfo_bb
    wl = make_workload(name)
    ctx = api.Context(0, ring="babybear")
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        assert (wit.w_ccs == wl.w_ccs).all()
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        tr = lambda: api.PoseidonTranscript(ring="babybear")
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
        inst = lfo_bb.Instance(wl)
        rc, lc_v = inst.verify(lfo_bb.Transcript(), acc, cccs, proof)
        assert rc == 0 and (lc_v == lc).all()
        cm0 = lc[wl.s + 9: wl.s + 9 + wl.kappa]
        assert (w0.commit(scheme) == cm0).all()
        ok, mx = ctx.linf_check(w0.f, wl.B // 2)
        assert ok, mx
        lc2, w02, proof2 = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
        assert (proof2 == proof).all()
        # chained step
        lc3, w3, proof3 = api.NIFSProver.prove(ctx, lc, w0, cccs, wit, tr())
        rc, lc_v = inst.verify(lfo_bb.Transcript(), lc, cccs, proof3)
        assert rc == 0 and (lc_v == lc3).all()
    finally:
        ctx.close()


def test_babybear_repeated_steps_are_stable():
    """10 chained BabyBear fold steps at B10: every step verifies (product verifier), norm stays below B/2, no device-memory growth"""
    wl = make_workload("B10")
    ctx = api.Context(0, ring="babybear")
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        tr = lambda: api.PoseidonTranscript(ring="babybear")

        def chain(steps):
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
            w = wit
            dig = []
            for i in range(steps):
                lc, w_next, proof = api.NIFSProver.prove(ctx, acc, w, cccs, wit, tr())
                ok, lc_v, stage = api.NIFSVerifier.verify(wl, acc, cccs, proof, tr())
                assert ok and (lc_v == lc).all(), (i, stage)
                good, mx = ctx.linf_check(w_next.f, wl.B // 2)
                assert good, (i, mx)
                dig.append(int(proof[-1, 0]) ^ int(lc[wl.s, 0]))
                if w is not wit:
                    w.free()
                acc, w = lc, w_next
            return dig

        d1 = chain(10)
        free1 = ctx.mem_info()[0]
        d2 = chain(10)
        free2 = ctx.mem_info()[0]
        assert d1 == d2
        assert abs(free1 - free2) < 64 << 20
    finally:
        ctx.close()


def test_contexts_are_thread_safe():
    """SURVEY 8(b) "Threading": reference callers may hit the ABI from several Rayon workers at once.  Four threads hammer
    one Goldilocks context and one BabyBear context concurrently (element-wise ops + commits); results must equal the
    single-threaded ones."""
    import threading
    g = api.Context(0)
    b = api.Context(0, ring="babybear")
    try:
        from latticefold_amd.workload import splitmix_fq
        xg = splitmix_fq(1, 0, 300 * 24).reshape(300, 24)
        xb = splitmix_fq(2, 0, 200 * 72, "babybear").reshape(200, 72)
        Ag = splitmix_fq(3, 0, 4 * 512 * 24).reshape(4, 512, 24)
        Ab = splitmix_fq(4, 0, 3 * 256 * 72, "babybear").reshape(3, 256, 72)
        sg = api.AjtaiCommitmentScheme(g, matrix=Ag)
        sb = api.AjtaiCommitmentScheme(b, matrix=Ab)
        fg = splitmix_fq(5, 0, 512 * 24).reshape(512, 24)
        fb = splitmix_fq(6, 0, 256 * 72, "babybear").reshape(256, 72)
        want = (g.crt(xg), g.icrt(xg), sg.commit_ntt(fg), b.crt(xb), b.icrt(xb), sb.commit_ntt(fb))
        errs = []

        def work(tid):
            try:
                for it in range(15):
                    got = (g.crt(xg), g.icrt(xg), sg.commit_ntt(fg), b.crt(xb), b.icrt(xb), sb.commit_ntt(fb))
                    for a, w in zip(got, want):
                        if not (a == w).all():
                            errs.append((tid, it))
            except Exception as e:  # noqa: BLE001
                errs.append((tid, repr(e)))

        ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errs, errs[:3]
    finally:
        g.close()
        b.close()
