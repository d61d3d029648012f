"""GPU parity tests of the BabyBearRingNTT backend (BASELINE configs[2]): every HIP path, called through the C ABI with
ring = LF_RING_BABYBEAR, must be BIT-EXACT against the BabyBear build of the CPU oracle (oracle/liblfo_bb.so) on the
same seeded inputs.  Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest

import lfo_bb as lfo
from latticefold_amd import api
from latticefold_amd.workload import diag, make_workload, splitmix_fq

pytestmark = pytest.mark.gpu
RING = "babybear"
P, RE, TAU = lfo.P, lfo.RE, lfo.TAU


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0, ring=RING)
    yield c
    c.close()


def rnd(seed, *shape):
    n = int(np.prod(shape))
    return splitmix_fq(seed, 0, n, RING).reshape(shape)


def tr_new():
    return api.PoseidonTranscript(ring=RING)


def setup_case(ctx, name, seed=0):
    wl = make_workload(name, seed)
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    return wl, inst, A, scheme


def test_device_arithmetic_selftest(ctx):
    """centred Montgomery words: F_p add/sub/mul and the 81-mad F_{p^9} product vs host % arithmetic, incl. edge operands"""
    for seed in (1, 2, 3):
        assert ctx.selftest_field(seed, 1 << 18) == 0


def test_ring_tables_match_oracle(ctx):
    nr, y = ctx.get_ring_tables()
    import ctypes as C
    onr = C.c_uint64()
    oy = np.zeros(8 * TAU, dtype=np.uint64)
    lfo.lib().lfo_get_ring(C.byref(onr), lfo._p64(oy))
    assert nr == onr.value and (y == oy).all()


def test_crt_with_other_ring_tables(ctx):
    """the CRT map is data for BabyBear too: permuted slots + another 9th root per slot must still match the oracle"""
    nr, y = ctx.get_ring_tables()
    y2 = y.reshape(8, TAU)[[3, 0, 6, 1, 7, 2, 5, 4]].copy()
    w3 = pow(31, (P - 1) // 3, P)           # a primitive cube root of unity: w3^9 = 1, so (w3 y)^9 = y^9
    assert w3 != 1
    y2 = np.array([[int(v) * pow(w3, k % 3, P) % P for v in row] for k, row in enumerate(y2)], dtype=np.uint64)
    try:
        ctx.set_ring_tables(nr, y2.reshape(-1))
        assert lfo.lib().lfo_set_ring(nr, lfo._p64(np.ascontiguousarray(y2.reshape(-1)))) == 0
        x = rnd(5, 300, RE)
        g = ctx.crt(x)
        assert (g == lfo.crt(x)).all()
        assert (ctx.icrt(g) == x).all()
        wl, inst, A, scheme = setup_case(ctx, "B6")      # and a whole fold step under the other map
        f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        acc_g, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr_new())
        acc_o, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
        lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
        lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
        assert (acc_g == acc_o).all() and (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
    finally:
        ctx.set_ring_tables(nr, y)
        lfo.lib().lfo_set_ring(nr, lfo._p64(np.ascontiguousarray(y)))
    with pytest.raises(api.LfError):
        bad = y.copy(); bad[1] = (int(bad[1]) + 1) % P
        ctx.set_ring_tables(nr, bad)


@pytest.mark.parametrize("count", [1, 7, 256, 1000])
def test_crt_icrt(ctx, count):
    x = rnd(11 + count, count, RE)
    assert (ctx.crt(x) == lfo.crt(x)).all()
    assert (ctx.icrt(x) == lfo.icrt(x)).all()
    assert (ctx.icrt(ctx.crt(x)) == x).all()


def test_decompose_recompose(ctx):
    x = rnd(3, 200, RE)
    x[0, :6] = [0, 1, P - 1, (P - 1) // 2, (P + 1) // 2, 2**15]
    for base, digits in ((1 << 16, 2), (1 << 8, 4), (2, 16)):
        src = x if base != 2 else lfo.decompose(x, 1 << 16, 2, 0)[:64]
        for layout in (0, 1):
            assert (ctx.decompose(src, base, digits, layout) == lfo.decompose(src, base, digits, layout)).all(), (base, layout)
        d = ctx.decompose(src, base, digits, 0)
        assert (ctx.recompose(d, base, digits) == src).all()
        assert (ctx.recompose(d, base, digits) == lfo.recompose(d, base, digits)).all()


def test_linf_check(ctx):
    c = np.zeros((50, RE), dtype=np.uint64)
    c[3, 5] = 70; c[9, 60] = P - 123
    f = lfo.crt(c)
    ok, mx = ctx.linf_check(f, 124)
    assert ok and mx == 123
    ok, mx = ctx.linf_check(f, 123)
    assert not ok
    ok, mx = ctx.linf_check(f, 1 << 20, unsigned_variant=True)
    assert not ok


@pytest.mark.parametrize("kappa,n,batch", [(5, 777, 3), (9, 2048, 1), (16, 1024, 15), (3, 64, 2), (16, 300, 20), (32, 70, 9)])
def test_ajtai_commit(ctx, kappa, n, batch):
    A = rnd(100 + kappa, kappa, n, RE)
    f = rnd(200 + n, batch, n, RE)
    s = api.AjtaiCommitmentScheme(ctx, matrix=A)
    got = s.commit_ntt(f)
    for b in range(batch):
        assert (got[b] == lfo.ajtai_commit(A, kappa, n, f[b])).all()
    with pytest.raises(api.CommitmentError):
        s.commit_ntt(f[0][:-1])


def test_ajtai_closed_form(ctx):
    """test_commit_ntt (commitment_scheme.rs:141-159) shape on BabyBear: diagonal scalars, closed form mod p"""
    kappa, n = 9, 1 << 12
    idx = (np.arange(kappa * n, dtype=np.uint64)).reshape(kappa, n)
    A = np.zeros((kappa, n, RE), dtype=np.uint64)
    A[:, :, 0::TAU] = idx[:, :, None]
    f = np.tile(diag(2, RING), (n, 1))
    got = api.AjtaiCommitmentScheme(ctx, matrix=A).commit_ntt(f)
    for i in range(kappa):
        assert (got[i] == diag(n * (2 * i * n + (n - 1)), RING)).all()


def test_ajtai_generate_matches_workload_stream(ctx):
    wl = make_workload("B8")
    A = wl.ajtai_matrix()
    f = rnd(9, wl.N, RE)
    dev = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
    assert (dev.commit_ntt(f) == lfo.ajtai_commit(A, wl.kappa, wl.N, f)).all()


def test_eq_and_mle_eval(ctx):
    for nv in (1, 5, 9):
        pt = rnd(nv, nv, TAU)
        ring_pt = np.tile(pt, (1, 8))
        eq = ctx.build_eq(pt)
        oeq = lfo.build_eq(ring_pt)
        assert (eq == oeq[:, 0:TAU]).all()
        for ln in ((1 << nv), max(1, (1 << nv) - 3)):
            tabs = rnd(50 + nv, 3, ln, RE)
            got = ctx.evaluate_mles(tabs, pt)
            for a in range(3):
                assert (got[a] == lfo.mle_eval(tabs[a], ring_pt)).all()
    with pytest.raises(api.LfError):
        ctx.evaluate_mles(rnd(1, 1, 40, RE), rnd(2, 5, TAU))


@pytest.mark.parametrize("name", ["B6", "BDP"])
def test_witness_roundtrips(ctx, name):
    wl, inst, A, scheme = setup_case(ctx, name)
    w = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    assert (w.f_coeff == f_coeff).all()
    assert (w.f == lfo.crt(f_coeff)).all()
    assert (w.w_ccs == wl.w_ccs).all()
    assert (w.commit(scheme) == lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_coeff))).all()
    assert (api.Witness.from_f(ctx, lfo.crt(f_coeff)).f_coeff == f_coeff).all()
    assert (api.Witness.from_f_coeff(ctx, f_coeff).f == w.f).all()
    big = f_coeff.copy(); big[0, 0] = wl.B
    with pytest.raises(api.LfError) as e:
        api.Witness.from_f_coeff(ctx, big)
    assert e.value.code == -5
    z = wl.z()
    for j in range(wl.t):
        got = ctx.mat_vec_mul(j, z)
        rows = min(wl.n, wl.m)
        exp = np.zeros((wl.m, RE), dtype=np.uint64)
        prod = np.zeros((rows, RE), dtype=np.uint64)
        lfo.lib().lfo_ring_mul_ntt(lfo._p64(np.ascontiguousarray(wl.val[j])), lfo._p64(np.ascontiguousarray(z[:rows])), lfo._p64(prod), rows)
        exp[:rows] = prod
        assert (got == exp).all()


def run_both(ctx, name, seed=0):
    wl, inst, A, scheme = setup_case(ctx, name, seed)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc_g, linpr_g = api.LFLinearizationProver.prove(ctx, cccs, wit, tr_new())
    acc_o, linpr_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    return wl, inst, A, f_coeff, wit, cccs, acc_g, linpr_g, acc_o, linpr_o


@pytest.mark.parametrize("name", ["B6", "BDP", "B8"])
def test_linearization_parity(ctx, name):
    wl, inst, A, f_coeff, wit, cccs, acc_g, linpr_g, acc_o, linpr_o = run_both(ctx, name)
    assert (linpr_g == linpr_o).all()
    assert (acc_g == acc_o).all()


@pytest.mark.parametrize("name", ["B6", "B10", "B14"])
def test_linearization_kernel_forms_agree(ctx, name, monkeypatch):
    """the linearization sumcheck through the R1CS kernel with fix_variables fused in (k_lin_r1cs; at most 256 pairs: message written by the round kernel itself),
    through the generic multiset kernel with the small rounds in one launch (k_lin_small) and through the generic kernel with separate fix / round / reduce
    launches: the same proof, the oracle's (nifs/linearization.rs:137-196, utils/sumcheck/prover.rs:56-162)"""
    if name == "B14":   # (the oracle is slow at this size: the three forms against each other; the default form is pinned by the scale digests)
        wl, inst, A, scheme = setup_case(ctx, name, 5)
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        acc_o, linpr_o = acc_g, linpr_g = api.LFLinearizationProver.prove(ctx, cccs, wit, tr_new())
    else:
        wl, inst, A, f_coeff, wit, cccs, acc_g, linpr_g, acc_o, linpr_o = run_both(ctx, name, 5)
    assert (linpr_g == linpr_o).all() and (acc_g == acc_o).all()
    for env in ({"LF_LIN_NO_R1CS": "1"}, {"LF_LIN_NO_R1CS": "1", "LF_LIN_NO_SMALL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        acc_e, linpr_e = api.LFLinearizationProver.prove(ctx, cccs, wit, tr_new())
        for k in env:
            monkeypatch.delenv(k)
        assert (linpr_e == linpr_o).all() and (acc_e == acc_o).all(), env


@pytest.mark.parametrize("name,seed", [("B6", 0), ("B6", 3), ("BDP", 0), ("B8", 1)])
def test_fold_step_parity(ctx, name, seed):
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, seed)
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    lin = wl.s * (wl.d + 2) + TAU + wl.t
    dec = wl.K * (wl.t + TAU + wl.l + 1 + wl.kappa)
    bad = np.nonzero((proof_g != proof_o).any(axis=1))[0]
    assert bad.size == 0, f"first differing proof elements {bad[:5]} (lin<{lin}, decL<{lin + dec}, decR<{lin + 2 * dec})"
    assert (lc_g == lc_o).all()
    assert (w0.f == f0_o).all()
    assert (w0.f_coeff == lfo.icrt(f0_o)).all()
    rc, lc_v = inst.verify(lfo.Transcript(), acc_g, cccs, proof_g)   # the restated NIFSVerifier accepts the GPU proof
    assert rc == 0 and (lc_v == lc_g).all()
    ok, lc_p, _ = api.NIFSVerifier.verify(wl, acc_g, cccs, proof_g, tr_new())   # ... and so does the product's own host verifier
    assert ok and (lc_p == lc_g).all()
    if name == "B6":   # chained step: the folded accumulator/witness are valid inputs of the next fold
        lc2_g, w2, proof2_g = api.NIFSProver.prove(ctx, lc_g, w0, cccs, wit, tr_new())
        lc2_o, f2_o, proof2_o = inst.fold_step(lfo.Transcript(), A, lc_o, lfo.icrt(f0_o), cccs, f_coeff)
        assert (proof2_g == proof2_o).all() and (lc2_g == lc2_o).all() and (w2.f == f2_o).all()


@pytest.mark.parametrize("name", ["B6"])
def test_fold_step_builds_f_and_w_ccs_of_the_folded_witness(ctx, name, monkeypatch):
    """Witness::from_f (arith.rs:299-313) builds f_coeff, f (NTT form) and w_ccs inside prove: a fold step materialises all three behind compute_f_0
    -- against the oracle's f_0 and its recomposition."""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    w_ccs_o = lfo.crt(lfo.recompose(lfo.icrt(f0_o), wl.B, wl.L))
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    assert (proof_g == proof_o).all() and (lc_g == lc_o).all()
    assert (w0.f == f0_o).all() and (w0.f_coeff == lfo.icrt(f0_o)).all() and (w0.w_ccs == w_ccs_o).all()
    w0.free()


@pytest.mark.parametrize("name", ["B8", "BDP"])
def test_fold_step_fused_fix_rounds_match_oracle(ctx, name, monkeypatch):
    """rounds >= 4 with fix_variables fused into the round kernel (threshold lowered to reach it at oracle sizes) and with the
    separate k_fix pass: identical proofs, equal to the oracle's"""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, 2)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    monkeypatch.setenv("LF_FOLD_FUSE_MIN", "4")
    monkeypatch.setenv("LF_FOLD_NO_LUT", "1")
    lc_f, w_f, proof_f = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    # rounds 3 and 4 from the 81-entry digit look-up table instead of materialised m/4-entry tables (large instances by default)
    monkeypatch.delenv("LF_FOLD_NO_LUT")
    monkeypatch.setenv("LF_FOLD_LUT_MIN", "1")
    lc_l, w_l, proof_l = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    # rounds 4 / 5 from the product-free tables over the digit codes (modes 6 / 7), round 5 on the stored round-4 tables, and the older modes 4 / 1
    for extra in ({"LF_FOLD_R5_MIN": "1"}, {"LF_FOLD_NO_R5TAB": "1"}, {"LF_FOLD_NO_R4TAB": "1"}):
        for key, val in extra.items():
            monkeypatch.setenv(key, val)
        lc_x, w_x, proof_x = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
        for key in extra:
            monkeypatch.delenv(key)
        assert (proof_x == proof_o).all() and (lc_x == lc_o).all() and (w_x.f == f0_o).all(), extra
    monkeypatch.delenv("LF_FOLD_LUT_MIN")
    monkeypatch.delenv("LF_FOLD_FUSE_MIN")
    monkeypatch.setenv("LF_FOLD_UNFUSED", "1")
    monkeypatch.setenv("LF_THETA_EVAL", "1")       # theta from evaluate_mles-style sums instead of the last fix of the sumcheck tables
    monkeypatch.setenv("LF_LIN_U_EVAL", "1")       # likewise u of the linearization
    lc_u, w_u, proof_u = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    assert (proof_f == proof_o).all() and (lc_f == lc_o).all() and (w_f.f == f0_o).all()
    assert (proof_l == proof_o).all() and (lc_l == lc_o).all() and (w_l.f == f0_o).all()
    assert (proof_u == proof_o).all() and (lc_u == lc_o).all()


@pytest.mark.parametrize("name", ["B10", "B8", "BDP"])
def test_fold_step_gemm_rounds_match_oracle(ctx, name, monkeypatch):
    """rounds 1..3 of the folding sumcheck as exact int8 GEMMs on the matrix cores (bb_sv_rounds.hip; from 16384 pairs on by default, LF_FOLD_SV_MIN lowers the
    threshold): one, two and three rounds in that form, followed by the look-up-table rounds, the fused-fix rounds or plain tables -- identical proofs, all
    equal to the oracle's (nifs/folding/utils.rs:273-325, utils/sumcheck/prover.rs:56-162)"""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, 2)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    monkeypatch.setenv("LF_FOLD_SV_MIN", "64")
    m = 1 << wl.s
    for rounds, extra in ((1, {}), (2, {}), (3, {}), (3, {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4"}), (2, {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4", "LF_FOLD_R5_MIN": "1"}),
                          (3, {"LF_FOLD_UNFUSED": "1"}), (3, {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4", "LF_FOLD_NO_R4TAB": "1"})):
        monkeypatch.setenv("LF_FOLD_SV_ROUNDS", str(rounds))
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        lc, w, proof = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
        for k in extra:
            monkeypatch.delenv(k)
        want = sum(1 << (r - 1) for r in range(1, rounds + 1) if (m >> r) >= 64) if wl.s >= 4 and wl.N <= m and wl.N % 4 == 0 else 0
        assert ctx.fold_paths() == want, (rounds, extra, ctx.fold_paths(), want)
        bad = np.nonzero((proof != proof_o).any(axis=1))[0]
        assert bad.size == 0 and (lc == lc_o).all() and (w.f == f0_o).all(), (rounds, extra, bad[:6])
    monkeypatch.setenv("LF_FOLD_NO_SV", "1")
    lc, w, proof = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    assert ctx.fold_paths() == 0 and (proof == proof_o).all()


@pytest.mark.parametrize("name", ["B10", "B8", "BDP"])
def test_fold_step_split_table_rounds_match_oracle(ctx, name, monkeypatch):
    """rounds 4 / 5 of the folding sumcheck in the split eq form (lfbb::k_fold_round SPLIT, modes 6 / 7: three lazy products per table, the G part from its own
    launch, the fourth coefficient from g(0) + g(1) = the previous message at its challenge on the host): the same words as the oracle's messages
    (nifs/folding/utils.rs:273-325, utils/sumcheck/prover.rs:56-162), after GEMM rounds and after the integer rounds, and identical with the form switched off"""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, 4)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    m = 1 << wl.s
    base = {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4", "LF_FOLD_SPLIT_MIN": "1"}
    # (round 5 from the planes: two lanes per pair, unsplit -- k_fold_round5_2l)
    cases = [({}, 0b01000), ({"LF_FOLD_R5_MIN": "1"}, 0b01000), ({"LF_FOLD_R5_MIN": "1", "LF_FOLD_SV_MIN": "64"}, 0b01000),
             ({"LF_FOLD_R5_MIN": "1", "LF_FOLD_ROUNDS_NO_SPLIT": "1"}, 0), ({"LF_FOLD_NO_R4TAB": "1"}, 0)]
    for extra, want in cases:
        env = dict(base, **extra)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        lc, w, proof = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
        for k in env:
            monkeypatch.delenv(k)
        if wl.s < 5 or m // 4 < 4:
            want = 0
        assert ctx.fold_split_rounds() == want, (extra, bin(ctx.fold_split_rounds()), bin(want))
        bad = np.nonzero((proof != proof_o).any(axis=1))[0]
        assert bad.size == 0 and (lc == lc_o).all() and (w.f == f0_o).all(), (extra, bad[:6])


@pytest.mark.parametrize("name", ["B6", "B10", "BDP", "BD768"])
def test_fold_step_int8_inner_products_match_oracle(ctx, name, monkeypatch):
    """u_s / eta as int8 GEMMs on the matrix cores (bb_dot_i8.hip; the driver uses them from 4096 columns on, LF_DOT_MIN lowers the
    threshold) and on the VALU kernel: identical proofs, equal to the oracle's"""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, 3)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    monkeypatch.setenv("LF_DOT_MIN", "64")
    lc_i, w_i, proof_i = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    monkeypatch.setenv("LF_DOT_VALU", "1")
    lc_v, w_v, proof_v = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    assert (proof_i == proof_o).all() and (lc_i == lc_o).all() and (w_i.f == f0_o).all()
    assert (proof_v == proof_o).all() and (lc_v == lc_o).all()


def test_sumcheck_lin_abi(ctx):
    wl, inst, A, scheme = setup_case(ctx, "B8")
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    cm = lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_coeff))
    cccs = np.concatenate([cm, wl.x_ccs])
    lc_o, pr_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    tr = tr_new()
    label = int.from_bytes(b"beta_s", "big") % P
    tr.absorb_slice(diag(label, RING)[None, :])
    beta = np.stack([tr.get_challenge() for _ in range(wl.s)])
    z = wl.z()
    tables = np.stack([ctx.mat_vec_mul(j, z) for j in range(wl.t)])
    sc = api.MLSumcheckLin(ctx, tables, beta)
    with pytest.raises(api.LfError) as e:
        sc.prove_round(r_prev=np.zeros(TAU, dtype=np.uint64))
    assert e.value.code == -7
    tr.absorb_slice(diag(wl.s, RING)[None, :]); tr.absorb_slice(diag(wl.d + 1, RING)[None, :])
    npts = wl.d + 2
    r = None
    for rd in range(wl.s):
        msg = sc.prove_round(r)
        assert (msg == pr_o[rd * npts:(rd + 1) * npts]).all(), rd
        tr.absorb_slice(msg)
        r = tr.get_challenge()
        ring_r = np.tile(r, 8)
        tr.absorb_slice(ring_r[None, :])
        assert (ring_r == lc_o[rd]).all()
    sc.end()


def test_ring_mismatch_errors(ctx):
    wl, inst, A, scheme = setup_case(ctx, "B6")
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    with pytest.raises(api.LfError) as e:      # a Goldilocks transcript cannot drive a BabyBear context
        api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
    assert e.value.code == -1
    with pytest.raises(api.LfError):
        ctx.set_sharding(0, 2, lambda x: np.stack([x, x]))


@pytest.mark.parametrize("name,ccs", [("B6", "deg3"), ("B6", "multi"), ("BDP", "deg3"), ("B8", "multi")])
def test_fold_step_parity_general_ccs(ctx, name, ccs):
    """arbitrary SparseMatrix rows and the degree-3 CCS of arith/ccs.rs:14-43 (t = 4, S = {{0,1,2},{3}})"""
    wl = make_workload(name, ccs=ccs)
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc_g, linpr_g = api.LFLinearizationProver.prove(ctx, cccs, wit, tr_new())
    acc_o, linpr_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    assert (linpr_g == linpr_o).all() and (acc_g == acc_o).all()
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    assert (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
    rc, lc_v = inst.verify(lfo.Transcript(), acc_g, cccs, proof_g)
    assert rc == 0 and (lc_v == lc_g).all()


def test_empty_and_degenerate_inputs(ctx):
    e = np.zeros((0, RE), dtype=np.uint64)
    assert ctx.crt(e).shape == (0, RE) and ctx.icrt(e).shape == (0, RE)
    one = rnd(77, 1, RE)
    assert (ctx.icrt(ctx.crt(one)) == one).all()
    assert (ctx.recompose(ctx.decompose(one, 1 << 16, 2, 0), 1 << 16, 2) == one).all()
    ok, mx = ctx.linf_check(lfo.crt(np.zeros((1, RE), dtype=np.uint64)), 1)
    assert ok and mx == 0
    A = rnd(5, 2, 1, RE)
    s = api.AjtaiCommitmentScheme(ctx, matrix=A)
    f = rnd(6, 1, RE)
    assert (s.commit_ntt(f) == lfo.ajtai_commit(A, 2, 1, f)).all()
    pt = rnd(8, 1, TAU)
    assert (ctx.evaluate_mles(rnd(9, 2, 1, RE), pt)[0] == lfo.mle_eval(rnd(9, 2, 1, RE)[0], np.tile(pt, (1, 8)))).all()


def test_fold_real_circuit_from_r1cs(ctx):
    """the reference's own test circuit x^3 + x + 5 = y (arith/r1cs.rs:128-151) through CCS::from_r1cs_padded; m = 8 rows (s = 3)"""
    from latticefold_amd import ccs
    A3, B3, C3 = ccs.vitalik_r1cs()
    z = ccs.vitalik_z_ntt(RING)
    wl = ccs.workload_from_r1cs(A3, B3, C3, 1, z[0:1], z[2:], ring=RING, L=2, Bbase=1 << 16, K=16, kappa=3, name="vitalik_bb")
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    assert (wit.w_ccs == wl.w_ccs).all()
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc_g, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr_new())
    acc_o, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    assert (acc_g == acc_o).all() and (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
    ok, lc_p, _ = api.NIFSVerifier.verify(wl, acc_g, cccs, proof_g, tr_new())
    assert ok and (lc_p == lc_g).all()


def test_fold_step_with_another_nonresidue(ctx, monkeypatch):
    """nu is data: F_{p^9} = F_p[Y]/(Y^9 - zeta) (zeta a primitive 24th root of unity instead of 2) switches the kernels to
    their generic-nu instantiation (Montgomery pre-multiplications instead of doublings)"""
    nr, y = ctx.get_ring_tables()
    g0 = 2
    while True:
        zeta = pow(g0, (P - 1) // 24, P)
        if pow(zeta, 12, P) != 1 and pow(zeta, 8, P) != 1:
            break
        g0 += 1
    E = [1, 5, 7, 11, 13, 17, 19, 23]
    y2 = np.zeros((8, TAU), dtype=np.uint64)
    for k, e in enumerate(E):
        g = e % 3                                     # (zeta^a Y^g)^9 = zeta^(9a + g) = zeta^e
        a = next(t for t in range(24) if (9 * t) % 24 == (e - g) % 24)
        y2[k, g] = pow(zeta, a, P)
    try:
        ctx.set_ring_tables(zeta, y2.reshape(-1))
        assert lfo.lib().lfo_set_ring(zeta, lfo._p64(np.ascontiguousarray(y2.reshape(-1)))) == 0
        x = rnd(5, 200, RE)
        assert (ctx.crt(x) == lfo.crt(x)).all() and (ctx.icrt(ctx.crt(x)) == x).all()
        assert ctx.selftest_field(7, 1 << 14) == 0
        wl, inst, A, f_coeff, wit, cccs, acc_g, linpr_g, acc_o, linpr_o = run_both(ctx, "B6")
        assert (linpr_g == linpr_o).all() and (acc_g == acc_o).all()
        lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
        lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
        assert (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
        rc, _ = inst.verify(lfo.Transcript(), acc_g, cccs, proof_g)
        assert rc == 0
        # the look-up-table / fused forms of rounds 3-6 in their generic-nu instantiations
        monkeypatch.setenv("LF_FOLD_LUT_MIN", "1")
        monkeypatch.setenv("LF_FOLD_FUSE_MIN", "4")
        lc_l, w_l, proof_l = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
        assert (proof_l == proof_o).all() and (lc_l == lc_o).all() and (w_l.f == f0_o).all()
        monkeypatch.setenv("LF_FOLD_R5_MIN", "1")      # ... and round 5 from the planes (mode 7)
        lc_l, w_l, proof_l = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
        assert (proof_l == proof_o).all() and (lc_l == lc_o).all() and (w_l.f == f0_o).all()
    finally:
        monkeypatch.delenv("LF_FOLD_R5_MIN", raising=False)
        monkeypatch.delenv("LF_FOLD_LUT_MIN", raising=False)
        monkeypatch.delenv("LF_FOLD_FUSE_MIN", raising=False)
        ctx.set_ring_tables(nr, y)
        lfo.lib().lfo_set_ring(nr, lfo._p64(np.ascontiguousarray(y)))


def test_fold_step_with_several_public_inputs(ctx):
    wl = make_workload("B6", l=3)
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc_g, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr_new())
    acc_o, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, tr_new())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    assert (acc_g == acc_o).all() and (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
    ok, lc_p, _ = api.NIFSVerifier.verify(wl, acc_g, cccs, proof_g, tr_new())
    assert ok and (lc_p == lc_g).all()
