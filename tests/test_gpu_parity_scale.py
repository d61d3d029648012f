"""Bit-exact GPU-vs-oracle parity AT THE BASELINE SIZES (the small-size word-for-word tests are tests/test_gpu_parity.py and
tests/test_gpu_bb.py).  Two kinds of check, all through the C ABI:

  * live oracle, complete fold step, word for word: C2 (BASELINE configs[1], 2^16 rows) -- `NIFSProver::prove`, nifs.rs:48-103;
  * committed oracle fixtures (tests/golden/scale_digests.json, produced by tests/tools/make_scale_digests.py with the oracle
    only): SHA-256 of every section of a complete fold step at C2, T18 (2^18), C4 (2^20, the metric config), B14 and C3 (BabyBear 2^18); general constraint
    systems (4 / 16 entries per row, the degree-three CCS) at C2 / B14 by digest and at C4 through the oracle's verifier, the commitment opening and the norm.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from latticefold_amd import api
from latticefold_amd.workload import make_workload

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scale_digests.json")


def _oracle(ring):
    if ring == "goldilocks":
        import lfo as O
    else:
        import lfo_bb as O
    return O


def _setup(name, ccs="r1cs"):
    wl = make_workload(name, ccs=ccs)
    ctx = api.Context(0, ring=wl.ring)
    ctx.load_ccs(wl)
    scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())   # same stream as wl.ajtai_matrix()
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    return wl, ctx, scheme, wit, cccs


def test_fold_step_bit_exact_vs_oracle_C2():
    wl, ctx, scheme, wit, cccs = _setup("C2")
    O = _oracle(wl.ring)
    try:
        tr = lambda: api.PoseidonTranscript(ring=wl.ring)
        acc, lin_pr = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
        inst = O.Instance(wl)
        A = inst.ajtai_matrix()
        f = inst.witness_from_w_ccs(wl.w_ccs)
        assert (wit.f_coeff == f).all()
        cm_o = O.ajtai_commit(A, wl.kappa, wl.N, O.crt(f))
        assert (cm_o == cccs[:wl.kappa]).all()
        acc_o, lin_o = inst.linearize(O.Transcript(), cccs, f)
        assert (acc == acc_o).all() and (lin_pr == lin_o).all()
        lc_o, f0_o, proof_o = inst.fold_step(O.Transcript(), A, acc_o, f, cccs, f)
        assert (proof == proof_o).all()
        assert (lc == lc_o).all()
        assert (w0.f == f0_o).all()
    finally:
        ctx.close()


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def _digests(wl, acc, lc, f0, proof):
    tau = wl.tau
    lin = wl.s * (wl.d + 2) + tau + wl.t
    dec = wl.K * (wl.t + tau + wl.l + 1 + wl.kappa)
    fm = wl.s * (2 * wl.b + 1)
    p = np.asarray(proof).reshape(-1, wl.RE)
    o = lin + 2 * dec
    return {"acc": _sha(acc), "lcccs_out": _sha(lc), "f0_ntt": _sha(f0), "proof_lin": _sha(p[:lin]),
            "proof_dec_left": _sha(p[lin:lin + dec]), "proof_dec_right": _sha(p[lin + dec:o]), "proof_fold_msgs": _sha(p[o:o + fm]),
            "proof_theta": _sha(p[o + fm:o + fm + 2 * wl.K * tau]), "proof_eta": _sha(p[o + fm + 2 * wl.K * tau:]), "proof": _sha(p)}


def _gold(name):
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/scale_digests.json missing (tests/tools/make_scale_digests.py)")
    g = json.load(open(GOLD))
    if name not in g:
        pytest.skip(f"no golden digest for {name}")
    return {k: v for k, v in g[name].items() if k.startswith(("acc", "lcccs", "f0", "proof"))}


@pytest.mark.parametrize("name", ["C2", "T18", "C4", "B14", "C3"])
def test_fold_step_matches_committed_oracle_digests(name):
    """complete fold step at the BASELINE sizes vs the committed oracle-only fixtures, section by section (linearization proof, both decompositions with their
    15 batched commits over the 5 GB matrix and 48 + 48 evaluations each, folding messages, theta, eta, the folded instance and witness); the live oracle runs
    next to the GPU at C2 above (a live C4 run of the oracle's linearization + decomposition cost the suite 30 s and compared the same words as the fixture)"""
    want = _gold(name)
    wl, ctx, scheme, wit, cccs = _setup(name)
    try:
        tr = lambda: api.PoseidonTranscript(ring=wl.ring)
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
        got = _digests(wl, acc, lc, w0.f, proof)
        bad = [k for k in want if got[k] != want[k]]
        assert not bad, f"{name}: sections differing from the oracle fixture: {bad}"
    finally:
        ctx.close()


@pytest.mark.parametrize("name,ccs", [("C2", "multi4"), ("C2", "multi16"), ("C2", "deg3"), ("B14", "multi4")])
def test_general_ccs_fold_step_matches_committed_oracle_digests(name, ccs):
    """SURVEY 8(f) row 2 at scale: general sparse matrices (4 / 16 entries per row at pseudo-random columns, ring-valued entries in C: arith/utils.rs:52-65 mat_vec_mul
    as a real CSR SpMV) and the reference's degree-three CCS (arith/ccs.rs:14-43, t = 4) at 2^16 rows -- complete fold steps vs oracle-only fixtures, section by section"""
    want = _gold(f"{name}/{ccs}")
    wl, ctx, scheme, wit, cccs = _setup(name, ccs)
    try:
        tr = lambda: api.PoseidonTranscript(ring=wl.ring)
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
        got = _digests(wl, acc, lc, w0.f, proof)
        bad = [k for k in want if got[k] != want[k]]
        assert not bad, f"{name}/{ccs}: sections differing from the oracle fixture: {bad}"
    finally:
        ctx.close()


@pytest.mark.parametrize("ccs", ["multi4", "multi16", "deg3"])
def test_general_ccs_fold_step_properties_at_c4(ccs):
    """the same constraint systems at the metric's size (2^20 rows; no oracle run fits there): the oracle's restated verifier accepts the proof and reproduces the folded
    LCCCS (O(proof) work plus M_j^T eq: the constraint matrices enter the check), the folded witness opens the folded commitment, its norm stays below B/2"""
    import lfo
    wl, ctx, scheme, wit, cccs = _setup("C4", ccs)
    try:
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, api.PoseidonTranscript())
        inst = lfo.Instance(wl)
        rc, lc_v = inst.verify(lfo.Transcript(), acc, cccs, proof)
        assert rc == 0 and (lc_v == lc).all()
        assert (w0.commit(scheme) == lc[wl.s + 3: wl.s + 3 + wl.kappa]).all()
        ok, mx = ctx.linf_check(w0.f, wl.B // 2)
        assert ok, mx
    finally:
        ctx.close()


@pytest.mark.parametrize("name,gemm_rounds", [("T14", False), ("B10", False), ("T14", True)])
def test_sub_provers_chain_to_fold_step(name, gemm_rounds, monkeypatch):
    """LFDecompositionProver / LFFoldingProver entry points (lf_decomposition_prove, lf_folding_prove): vs the oracle's
    decomposition, and chained after the linearization they reproduce NIFSProver::prove (nifs.rs:59-103) bit for bit.  gemm_rounds: the
    stand-alone folding prover with rounds 1-3 of its sumcheck as int8 GEMMs (it builds the bit planes of the witnesses itself)"""
    if gemm_rounds:
        monkeypatch.setenv("LF_FOLD_SV_MIN", "64")
    wl, ctx, scheme, wit, cccs = _setup(name)
    O = _oracle(wl.ring)
    try:
        tr = lambda: api.PoseidonTranscript(ring=wl.ring)
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
        inst = O.Instance(wl)
        A = wl.ajtai_matrix()
        f = inst.witness_from_w_ccs(wl.w_ccs)
        lcs, dec_pr = api.LFDecompositionProver.prove(ctx, acc, wit, tr())
        lcs_o, dec_o = inst.decomposition_prove(O.Transcript(), A, acc, f)
        assert (dec_pr == dec_o).all() and (lcs == lcs_o).all()
        # replay nifs.rs:59-103 with the three sub-provers on one transcript
        t1 = tr()
        lbl = lambda s: np.array([int.from_bytes(s.encode(), "big") % wl.P], dtype=np.uint64)
        from latticefold_amd.workload import diag
        t1.absorb_slice(diag(int(lbl("acc")[0]), wl.ring)); t1.absorb_slice(acc)
        t1.absorb_slice(diag(int(lbl("cm_i")[0]), wl.ring)); t1.absorb_slice(cccs)
        lin_lc, lin_pr = api.LFLinearizationProver.prove(ctx, cccs, wit, t1)
        lcs_l, dec_l = api.LFDecompositionProver.prove(ctx, acc, wit, t1)
        lcs_r, dec_r = api.LFDecompositionProver.prove(ctx, lin_lc, wit, t1)
        lc2, w02, fold_pr = api.LFFoldingProver.prove(ctx, np.concatenate([lcs_l, lcs_r]), wit, wit, t1)
        assert (np.concatenate([lin_pr, dec_l, dec_r, fold_pr]) == proof).all()
        assert (lc2 == lc).all() and (w02.f == w0.f).all()
        if wl.ring == "goldilocks":
            assert ctx.fold_paths() == (7 if gemm_rounds else 0)
    finally:
        ctx.close()
