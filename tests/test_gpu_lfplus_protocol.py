"""LatticeFold+ monomial set check and range check on the GPU (lfplus_set_check, lfplus_range_check; src/setchk.rs:65-262, src/rgchk.rs:81-186)
against the oracle (oracle/lfp_protocol.c), word for word: sumcheck messages, the point, every evaluation; the transcripts end in the same
state; the product's host verifier and the oracle's verifier accept the GPU proofs."""
import numpy as np
import pytest

import lfp
from latticefold_amd import plus

pytestmark = pytest.mark.gpu
P, D = plus.P, plus.D


@pytest.fixture(scope="module")
def ctx():
    c = plus.PlusContext(0)
    yield c
    c.close()


def _ident(n, first=None):
    rowptr, col = np.arange(n + 1, dtype=np.uint32), np.arange(n, dtype=np.uint32)
    val = np.zeros((n, D), dtype=np.uint64)
    val[:, 0] = 1
    if first is not None:
        val[0, 0] = first
    return rowptr, col, val


def _rand_csr(n, seed, per_row=2):
    rng = np.random.default_rng(seed)
    rowptr = np.arange(0, per_row * n + 1, per_row, dtype=np.uint32)
    col = rng.integers(0, n, size=per_row * n).astype(np.uint32)
    val = lfp.splitmix(seed, 0, per_row * n * D).reshape(-1, D)
    return rowptr, col, val


@pytest.mark.parametrize("nvars,nmat,ncols,nvec,nM", [(2, 1, 4, 0, 0), (2, 2, 4, 0, 0), (3, 2, 4, 2, 0), (5, 1, 16, 1, 1), (10, 3, 16, 2, 2), (13, 2, 16, 1, 1)])
@pytest.mark.parametrize("round0_ahead", [True, False])
def test_set_check_matches_oracle(ctx, nvars, nmat, ncols, nvec, nM, round0_ahead, monkeypatch):
    """shapes of setchk.rs:355-495 (one / batched / mixed sets) and of the range check (16 columns, vectors, M_i rows); random exponents,
    a few absent entries (identity-like sparse sets).  round0_ahead: every set's share of round 0 is computed behind its tables and the message combined with
    the batching challenge on the host (the default with more than one matrix set); False (LFPLUS_SC_NO_EARLY=1): round 0 in the loop, over all the sets"""
    if round0_ahead:
        monkeypatch.delenv("LFPLUS_SC_NO_EARLY", raising=False)
    else:
        monkeypatch.setenv("LFPLUS_SC_NO_EARLY", "1")
    n = 1 << nvars
    rng = np.random.default_rng(nvars * 100 + nmat)
    dig = rng.integers(-7, 8, size=(nmat, n, ncols)).astype(np.int8)
    dense = lfp.exp_dense(dig)
    if nvars <= 3:                        # sparse sets: absent entries are zero ring elements
        mask = rng.random(dig.shape) < 0.4
        dig[mask] = plus.ABSENT
        dense[mask] = 0
    vdig = rng.integers(-7, 8, size=(nvec, n)).astype(np.int8) if nvec else None
    mats = [_ident(n, first=5)] + [_rand_csr(n, 7 + q) for q in range(1, nM)] if nM else []
    to, tp = lfp.Transcript(), plus.PoseidonTranscript()
    want = lfp.set_check(to, nvars, dense, lfp.exp_dense(vdig) if nvec else None, mats)
    got = plus.set_check(ctx, tp, nvars, dig, vdig, mats)
    for key in ("msgs", "r", "e", "b"):
        assert (got[key] == want[key]).all(), key
    assert tp.get_challenge() == to.challenge()            # the transcripts end in the same state
    ok, stage, r = plus.set_check_verify(plus.PoseidonTranscript(), nvars, got, nM=nM)
    rc_o = lfp.set_check_verify(lfp.Transcript(), nvars, got, nM=nM)[0]
    if nmat == 1 and nvec:
        # a quirk of the reference, restated literally: without a batching challenge (a single matrix set) the prover's closure returns after
        # the first matrix set (setchk.rs:172-176), so vector sets never enter the sumcheck, while Out::verify adds their claims (setchk.rs:318-333)
        assert not ok and stage == 3 and rc_o == -3
    else:
        assert ok and (r == got["r"]).all() and rc_o == 0


@pytest.mark.parametrize("nvars,nmat,nvec,nM", [(3, 2, 0, 0), (4, 3, 2, 0), (6, 2, 1, 1), (11, 4, 2, 2)])
def test_set_check_digit_rounds_equal_table_rounds(ctx, nvars, nmat, nvec, nM, monkeypatch):
    """Rounds 0 and 1 of the set check read the exponent digits (k_sc_round0_dig, k_sc_fix_round_dig: 16-column matrix sets and vector sets, absent entries as
    the zero monomial); LFPLUS_SC_TABLES=1 (read per call) materialises the beta^e tables first.  Both forms against the oracle, word for word."""
    n = 1 << nvars
    rng = np.random.default_rng(nvars * 7 + nmat)
    dig = rng.integers(-7, 8, size=(nmat, n, 16)).astype(np.int8)
    dense = lfp.exp_dense(dig)
    mask = rng.random(dig.shape) < 0.3
    dig[mask] = plus.ABSENT
    dense[mask] = 0
    vdig = rng.integers(-7, 8, size=(nvec, n)).astype(np.int8) if nvec else None
    mats = [_ident(n, first=5)] + [_rand_csr(n, 7 + q) for q in range(1, nM)] if nM else []
    want = lfp.set_check(lfp.Transcript(), nvars, dense, lfp.exp_dense(vdig) if nvec else None, mats)
    for tables in (False, True):
        if tables:
            monkeypatch.setenv("LFPLUS_SC_TABLES", "1")
        else:
            monkeypatch.delenv("LFPLUS_SC_TABLES", raising=False)
        got = plus.set_check(ctx, plus.PoseidonTranscript(), nvars, dig, vdig, mats)
        for key in ("msgs", "r", "e", "b"):
            assert (got[key] == want[key]).all(), (key, tables)


def _witness(n, seed):
    v = (lfp.splitmix(seed, 0, n * D) % np.uint64(63)).astype(np.int64) - 31
    return np.where(v < 0, np.uint64(P) - (-v).astype(np.uint64), v.astype(np.uint64)).reshape(n, D)


@pytest.mark.parametrize("L,nM,nvars,ring_weights", [(1, 0, 14, False), (1, 1, 14, False), (2, 1, 14, False), (1, 1, 15, False), (2, 1, 14, True)])
def test_range_check_matches_oracle(L, nM, nvars, ring_weights, monkeypatch):
    """rgchk.rs:340-433 (test_range_check: n = 2^15, kappa 1, k 2; test_range_check_mm: one matrix) and two instances as Mlin::mlin builds them:
    RgInstance::from_f on the GPU, then the range check on the resident instances.  ring_weights: w = M^T eq(r) kept as ring elements although the matrix has
    constant coefficients (LFPLUS_RING_WEIGHTS=1: the form matrices with ring coefficients take) -- the same evaluations as the scalar weights of the default"""
    if ring_weights:
        monkeypatch.setenv("LFPLUS_RING_WEIGHTS", "1")
    else:
        monkeypatch.delenv("LFPLUS_RING_WEIGHTS", raising=False)
    n, kappa, k = 1 << nvars, 1, 2
    dp = plus.DecompParameters.for_frog(k)
    A = lfp.splitmix(1, 0, kappa * n * D).reshape(kappa, n, D)
    ctxs, insts = [], []
    try:
        for l in range(L):
            f = _witness(n, 20 + l)
            c = plus.PlusContext(0)
            rg = plus.RgInstance.from_f(c, f, A, dp)
            ctxs.append(c)
            insts.append({"Mf": lfp.exp_dense(rg.D_f), "tau": rg.tau, "mtau": lfp.exp_dense(rg.m_tau_exp), "f": f})
        mats = [_ident(n, first=2)] * nM
        to, tp = lfp.Transcript(), plus.PoseidonTranscript()
        want = lfp.range_check(to, nvars, insts, k, mats)
        got = plus.range_check(ctxs, tp, mats)
        for key in ("msgs", "r", "e", "b", "v", "a", "bb", "c"):
            assert (got[key] == want[key]).all(), key
        assert tp.get_challenge() == to.challenge()
        ok, stage, r = plus.range_check_verify(plus.PoseidonTranscript(), got)
        assert ok and stage == 0
        assert lfp.range_check_verify(lfp.Transcript(), nvars, got, k)[0] == 0
    finally:
        for c in ctxs:
            c.close()


def test_argument_errors(ctx):
    tp = plus.PoseidonTranscript()
    dig = np.zeros((1, 4, 4), dtype=np.int8)
    dig[0, 0, 0] = 8                      # outside (-8, 8): the reference's exp() returns None
    with pytest.raises(plus.LfPlusError) as e:
        plus.set_check(ctx, tp, 2, dig)
    assert e.value.code == plus.E_EXP_DOMAIN
    c2 = plus.PlusContext(0)
    try:
        with pytest.raises(plus.LfPlusError) as e:
            plus.range_check([c2], tp)      # no resident RgInstance
        assert e.value.code == plus.E_ARG
    finally:
        c2.close()


@pytest.mark.parametrize("L,nM,kappa,nvars,ring_coeffs", [(1, 0, 1, 14, False), (1, 1, 2, 15, False), (2, 1, 1, 14, False), (2, 2, 2, 15, False), (2, 2, 2, 15, True),
                                                         (3, 1, 3, 16, False), (2, 0, 4, 16, False)])   # three instances, kappa not a power of two (tensor(c) has 4 entries, comh 3)
@pytest.mark.parametrize("all_tables", [False, True])
def test_cm_prove_matches_oracle(L, nM, kappa, nvars, ring_coeffs, all_tables, monkeypatch):
    """cm.rs:606-666 (test_com: n = 2^15, kappa 2, k 2, one matrix) and the two-instance shape Mlin::mlin feeds Cm::prove: every proof field, the
    folded instance and the folded witness equal the oracle's; both verifiers accept; the transcripts end in the same state.
    all_tables: the sumcheckers' rounds over every instance table (LFPLUS_CM_FULL=1, the reference's shape) instead of the batched tables eq U + V Z with the
    evaluations taken from the original tables (the default) -- the same proof either way"""
    if all_tables:
        monkeypatch.setenv("LFPLUS_CM_FULL", "1")
    else:
        monkeypatch.delenv("LFPLUS_CM_FULL", raising=False)
    monkeypatch.delenv("LFPLUS_CM_UNFUSED", raising=False)
    _cm_prove_case(L, nM, kappa, nvars, ring_coeffs)


@pytest.mark.parametrize("L,nM,kappa,nvars", [(2, 1, 1, 14), (2, 2, 2, 15), (3, 1, 3, 16)])
@pytest.mark.parametrize("all_tables", [False, True])
def test_cm_prove_unfused_fix_matches_oracle(L, nM, kappa, nvars, all_tables, monkeypatch):
    """The sumcheckers of Cm::prove defer fix_variables into the next round's kernel (k_cm_round_fused / k_cm2_round<true>); LFPLUS_CM_UNFUSED=1 runs the separate
    k_cm_fix passes instead.  The switch is read on every call, so both forms are compared with the oracle here (the default form is the test above)."""
    monkeypatch.setenv("LFPLUS_CM_UNFUSED", "1")
    if all_tables:
        monkeypatch.setenv("LFPLUS_CM_FULL", "1")
    else:
        monkeypatch.delenv("LFPLUS_CM_FULL", raising=False)
    _cm_prove_case(L, nM, kappa, nvars, False)


@pytest.mark.parametrize("L,nM,kappa,nvars", [(1, 1, 2, 15), (2, 2, 2, 15), (3, 1, 3, 16), (3, 3, 2, 15)])
def test_cm_prove_dense_tables_match_oracle(L, nM, kappa, nvars, monkeypatch):
    """By default the batched sumcheckers read m_tau as exponent bytes and the M_q tau as scalars when every M_q has constant coefficients (k_cm_combine_c,
    launch_cm_evals_c; the default form is test_cm_prove_matches_oracle above); LFPLUS_CM_DENSE=1 (read per call) materialises every table as ring elements --
    the form a sharded prove and shapes beyond the compact form's limits (L > 8, more than 64 tables) run."""
    monkeypatch.setenv("LFPLUS_CM_DENSE", "1")
    monkeypatch.delenv("LFPLUS_CM_FULL", raising=False)
    monkeypatch.delenv("LFPLUS_CM_UNFUSED", raising=False)
    _cm_prove_case(L, nM, kappa, nvars, False)


def test_cm_prove_compact_tables_three_matrices(monkeypatch):
    """the bench's shape in small: three instances, three constant-coefficient matrices (12 of 47 tables compact)"""
    for key in ("LFPLUS_CM_DENSE", "LFPLUS_CM_FULL", "LFPLUS_CM_UNFUSED"):
        monkeypatch.delenv(key, raising=False)
    _cm_prove_case(3, 3, 2, 15, False)


_CM_ORACLE = {}   # (case) -> (oracle proof, oracle transcript's next challenge): the switch variants of one case compare with the same oracle run


def _cm_prove_case(L, nM, kappa, nvars, ring_coeffs):
    n, k = 1 << nvars, 2
    dp = plus.DecompParameters.for_frog(k)
    A = lfp.splitmix(3, 0, kappa * n * D).reshape(kappa, n, D)
    ctxs, insts = [], []
    try:
        for l in range(L):
            f = _witness(n, 40 + l)
            c = plus.PlusContext(0)
            rg = plus.RgInstance.from_f(c, f, A, dp)
            ctxs.append(c)
            insts.append({"Mf": lfp.exp_dense(rg.D_f), "tau": rg.tau, "mtau": lfp.exp_dense(rg.m_tau_exp), "f": f, "comMf": rg.comM_f,
                          "fcoms": np.stack([rg.fcoms.cm_f, rg.fcoms.C_Mf, rg.fcoms.cm_mtau])})
        rnd = _rand_csr(n, 9)
        if not ring_coeffs:
            rnd[2][:, 1:] = 0          # constant coefficients: ct(psi M exp(tau)) = M tau needs them (the reference's tests use the identity)
        mats = [_ident(n, first=2), rnd][:nM]
        tp = plus.PoseidonTranscript()
        case = (L, nM, kappa, nvars, ring_coeffs)
        if case not in _CM_ORACLE:
            to = lfp.Transcript()
            # the instances are the product's from_f outputs (themselves checked against the oracle in test_range_check_* / test_gpu_lfplus.py); pin them so that a
            # cached oracle run is only reused for identical inputs
            pin = [hash(i[key].tobytes()) for i in insts for key in ("Mf", "tau", "mtau", "f")]
            want = lfp.cm_prove(to, nvars, insts, k, dp.l, kappa, mats)
            _CM_ORACLE[case] = (want, to.challenge(), pin)
        want, want_chal, pin = _CM_ORACLE[case]
        assert pin == [hash(i[key].tobytes()) for i in insts for key in ("Mf", "tau", "mtau", "f")]
        got = plus.cm_prove(ctxs, tp, dp.l, mats, want_g=True)
        for key in ("msgs", "r", "e", "b", "v", "a", "bb", "c", "comh", "pa", "ea", "pb", "eb", "ro", "cm_g", "vo", "g"):
            assert (got[key] == want[key]).all(), key
        assert tp.get_challenge() == want_chal
        for l in range(L):
            assert (plus.cm_read_g(ctxs[l]) == want["g"][l]).all()
        fcoms = [i["fcoms"] for i in insts]
        ok, stage, x = plus.cm_verify(plus.PoseidonTranscript(), got, fcoms)
        if ring_coeffs:                # an honest rejection: the range relation does not survive non-constant matrix coefficients
            assert not ok and stage == 4 and lfp.cm_verify(lfp.Transcript(), got, fcoms)[0] == -4
        else:
            assert ok and stage == 0 and all((x[key] == got[key]).all() for key in ("cm_g", "ro", "vo"))
            assert lfp.cm_verify(lfp.Transcript(), got, fcoms)[0] == 0
    finally:
        for c in ctxs:
            c.close()
