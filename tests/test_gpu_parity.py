"""GPU parity tests: every HIP path, called through the C ABI, must be BIT-EXACT against the CPU oracle
(oracle/) on the same seeded inputs.  Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest

import lfo
from latticefold_amd import api
from latticefold_amd.workload import P, RE, diag, make_workload, splitmix_fq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def rnd(seed, *shape):
    n = int(np.prod(shape))
    return splitmix_fq(seed, 0, n).reshape(shape)


def setup_case(ctx, name, seed=0):
    wl = make_workload(name, seed)
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    return wl, inst, A, scheme


def test_device_arithmetic_selftest(ctx):
    """fast NU=2^40 F_{p^3} product, (L,H) lazy sums and partial-product accumulators vs the generic path, 4M operand sets"""
    for seed in (1, 2, 3, 4):
        assert ctx.selftest_field(seed, 1 << 20) == 0


# ---- element-wise kernels ------------------------------------------------------------------------------
@pytest.mark.parametrize("count", [1, 7, 256, 1000])
def test_crt_icrt(ctx, count):
    x = rnd(11 + count, count, RE)
    assert (ctx.crt(x) == lfo.crt(x)).all()
    assert (ctx.icrt(x) == lfo.icrt(x)).all()
    assert (ctx.icrt(ctx.crt(x)) == x).all()


def test_crt_with_other_ring_tables(ctx):
    """the CRT map is data: permuted slots + a different cube root per slot must still match the oracle"""
    nr, y = ctx.get_ring_tables()
    y2 = y.reshape(8, 3)[[3, 0, 6, 1, 7, 2, 5, 4]].copy()
    # multiply every y_k by a primitive cube root of unity in F_p: still a cube root of the same zeta
    w3 = pow(7, (P - 1) // 3, P)
    y2 = np.array([[int(v) * pow(w3, k % 3, P) % P for v in row] for k, row in enumerate(y2)], dtype=np.uint64)
    try:
        ctx.set_ring_tables(nr, y2.reshape(-1))
        assert lfo.lib().lfo_set_ring(nr, lfo._p64(np.ascontiguousarray(y2.reshape(-1)))) == 0
        x = rnd(5, 300, RE)
        g = ctx.crt(x)
        assert (g == lfo.crt(x)).all()
        assert (ctx.icrt(g) == x).all()
    finally:
        ctx.set_ring_tables(nr, y)
        lfo.lib().lfo_set_ring(nr, lfo._p64(np.ascontiguousarray(y)))
    with pytest.raises(api.LfError):
        bad = y.copy(); bad[1] = (int(bad[1]) + 1) % P
        ctx.set_ring_tables(nr, bad)


def test_decompose_recompose(ctx):
    x = rnd(3, 300, RE)
    x[0, :6] = [0, 1, P - 1, (P - 1) // 2, (P + 1) // 2, 2**15]
    for base, digits in ((1 << 16, 4), (1 << 15, 5), (2, 16)):
        src = x if base != 2 else lfo.decompose(x, 1 << 16, 4, 0)[:64]
        for layout in (0, 1):
            g = ctx.decompose(src, base, digits, layout)
            o = lfo.decompose(src, base, digits, layout)
            assert (g == o).all(), (base, layout)
        d = ctx.decompose(src, base, digits, 0)
        assert (ctx.recompose(d, base, digits) == src).all()
        assert (ctx.recompose(d, base, digits) == lfo.recompose(d, base, digits)).all()
    with pytest.raises(api.LfError):
        ctx.decompose(x, 10485760000, 8, 0)   # StarkDP base: unsupported (non power of two)


def test_linf_check(ctx):
    c = np.zeros((50, RE), dtype=np.uint64)
    c[3, 5] = 70; c[9, 0] = P - 123
    f = lfo.crt(c)
    ok, mx = ctx.linf_check(f, 124)
    assert ok and mx == 123
    ok, mx = ctx.linf_check(f, 123)
    assert not ok
    ok, mx = ctx.linf_check(f, 1 << 20, unsigned_variant=True)   # literal Witness::within_bound: p-123 >= bound
    assert not ok


# kappa*batch in [384, 448] takes the SIMD-balanced 8-wave layout of k_ajtai (two lanes per output on waves 4-7, dot-product
# kernel for the outputs beyond 384); the other shapes the one-thread-per-output map
@pytest.mark.parametrize("kappa,n,batch", [(5, 777, 3), (9, 4096, 1), (26, 1024, 15), (3, 64, 2), (26, 512, 40),
                                           (24, 333, 16), (28, 97, 16), (25, 650, 16), (48, 70, 9),    # 48 rows of A: one LDS tile (57 rows, 88 KB of the 160 KB)
                                           (99, 200, 31), (49, 130, 5), (128, 64, 2)])                   # kappa > 48: equal row chunks (3 x 33, 2 x 25 / 24, 3 x 43 / 42) + scatter
def test_ajtai_commit(ctx, kappa, n, batch):
    A = rnd(100 + kappa, kappa, n, RE)
    f = rnd(200 + n, batch, n, RE)
    s = api.AjtaiCommitmentScheme(ctx, matrix=A)
    got = s.commit_ntt(f)
    for b in range(batch):
        assert (got[b] == lfo.ajtai_commit(A, kappa, n, f[b])).all()
    with pytest.raises(api.CommitmentError):
        s.commit_ntt(f[0][:-1])


def test_ajtai_closed_form_kat(ctx, kats):
    """test_commit_ntt (commitment_scheme.rs:141-159) at its real size N = 2^15, kappa = 9"""
    k = kats["commit_ntt"]
    kappa, n = k["kappa"], k["n"]
    idx = (np.arange(kappa * n, dtype=np.uint64)).reshape(kappa, n)
    A = np.zeros((kappa, n, RE), dtype=np.uint64)
    A[:, :, 0::3] = idx[:, :, None]
    f = np.tile(diag(2), (n, 1))
    got = api.AjtaiCommitmentScheme(ctx, matrix=A).commit_ntt(f)
    for i in range(kappa):
        assert (got[i] == diag(n * (2 * i * n + (n - 1)))).all()


def test_ajtai_generate_matches_workload_stream(ctx):
    wl = make_workload("T8")
    A = wl.ajtai_matrix()
    f = rnd(9, wl.N, RE)
    dev = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
    assert (dev.commit_ntt(f) == lfo.ajtai_commit(A, wl.kappa, wl.N, f)).all()


def test_eq_and_mle_eval(ctx):
    for nv in (1, 5, 6, 7, 10, 13):   # nv >= 6 takes the two-level (outer product) build, odd nv splits unevenly
        pt = rnd(nv, nv, 3)
        ring_pt = np.zeros((nv, RE), dtype=np.uint64)
        for k in range(8):
            ring_pt[:, 3 * k:3 * k + 3] = pt
        eq = ctx.build_eq(pt)
        oeq = lfo.build_eq(ring_pt)
        assert (eq == oeq[:, 0:3]).all()
        for ln in ((1 << nv), max(1, (1 << nv) - 3)):
            tabs = rnd(50 + nv, 3, ln, RE)
            got = ctx.evaluate_mles(tabs, pt)
            for a in range(3):
                assert (got[a] == lfo.mle_eval(tabs[a], ring_pt)).all()
    with pytest.raises(api.LfError):
        ctx.evaluate_mles(rnd(1, 1, 40, RE), rnd(2, 5, 3))   # IncorrectLength: 40 > 2^5


# ---- witness plumbing ------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["T8", "G5"])
def test_witness_roundtrips(ctx, name):
    wl, inst, A, scheme = setup_case(ctx, name)
    w = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    assert (w.f_coeff == f_coeff).all()
    assert (w.f == lfo.crt(f_coeff)).all()
    assert (w.w_ccs == wl.w_ccs).all()            # test_from_w_ccs (arith.rs:516-526)
    assert (w.commit(scheme) == lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_coeff))).all()
    w2 = api.Witness.from_f(ctx, lfo.crt(f_coeff))
    assert (w2.f_coeff == f_coeff).all()           # test_from_f
    w3 = api.Witness.from_f_coeff(ctx, f_coeff)
    assert (w3.f == w.f).all()                     # test_from_f_coeff
    big = f_coeff.copy(); big[0, 0] = wl.B
    with pytest.raises(api.LfError) as e:
        api.Witness.from_f_coeff(ctx, big)
    assert e.value.code == -5                       # LF_ERR_NORM
    z = wl.z()
    for j in range(wl.t):
        got = ctx.mat_vec_mul(j, z)
        # oracle SpMV through a 1-table MLE evaluation at hypercube points is slow; check rows directly
        exp = np.zeros((wl.m, RE), dtype=np.uint64)
        rows = min(wl.n, wl.m)
        prod = np.zeros((rows, RE), dtype=np.uint64)
        lfo.lib().lfo_ring_mul_ntt(lfo._p64(np.ascontiguousarray(wl.val[j])), lfo._p64(np.ascontiguousarray(z[:rows])), lfo._p64(prod), rows)
        exp[:rows] = prod
        assert (got == exp).all()


# ---- protocol ----------------------------------------------------------------------------------------------
def run_both(ctx, name, seed=0):
    wl, inst, A, scheme = setup_case(ctx, name, seed)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cm = wit.commit(scheme)
    cccs = np.concatenate([cm, wl.x_ccs])
    acc_g, linpr_g = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
    acc_o, linpr_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    return wl, inst, A, f_coeff, wit, cccs, acc_g, linpr_g, acc_o, linpr_o


@pytest.mark.parametrize("name", ["T8", "G5", "T10"])
def test_linearization_parity(ctx, name):
    wl, inst, A, f_coeff, wit, cccs, acc_g, linpr_g, acc_o, linpr_o = run_both(ctx, name)
    assert (linpr_g == linpr_o).all()
    assert (acc_g == acc_o).all()


@pytest.mark.parametrize("name,seed", [("T8", 0), ("T8", 3), ("G5", 0), ("T10", 1)])
def test_fold_step_parity(ctx, name, seed):
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, seed)
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    sizes = dict(lin=wl.s * (wl.d + 2) + 3 + wl.t, dec=wl.K * (wl.t + 3 + wl.l + 1 + wl.kappa))
    bad = np.nonzero((proof_g != proof_o).any(axis=1))[0]
    assert bad.size == 0, f"first differing proof element {bad[:5]} (lin<{sizes['lin']}, decL<{sizes['lin'] + sizes['dec']})"
    assert (lc_g == lc_o).all()
    assert (w0.f == f0_o).all()
    assert (w0.f_coeff == lfo.icrt(f0_o)).all()
    # the restated NIFSVerifier accepts the GPU proof (nifs/tests.rs:58-117 style)
    rc, lc_v = inst.verify(lfo.Transcript(), acc_g, cccs, proof_g)
    assert rc == 0 and (lc_v == lc_g).all()
    ok, lc_p, _ = api.NIFSVerifier.verify(wl, acc_g, cccs, proof_g, api.PoseidonTranscript())   # ... and so does the product's own host verifier
    assert ok and (lc_p == lc_g).all()
    assert (api.proof_from_bytes(wl, api.proof_to_bytes(wl, proof_g)) == proof_g).all()   # wire format round trip of a GPU proof
    # fold again: the folded accumulator/witness are valid inputs of the next step (IVC chaining)
    lc2_g, w2, proof2_g = api.NIFSProver.prove(ctx, lc_g, w0, cccs, wit, api.PoseidonTranscript())
    lc2_o, f2_o, proof2_o = inst.fold_step(lfo.Transcript(), A, lc_o, lfo.icrt(f0_o), cccs, f_coeff)
    assert (proof2_g == proof2_o).all() and (lc2_g == lc2_o).all() and (w2.f == f2_o).all()


@pytest.mark.parametrize("name", ["T8"])
def test_fold_step_builds_f_and_w_ccs_of_the_folded_witness(ctx, name, monkeypatch):
    """Witness::from_f (arith.rs:299-313) builds f_coeff, f (NTT form) and w_ccs inside prove: a fold step materialises all three behind compute_f_0
    -- against the oracle's f_0 and its recomposition."""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    w_ccs_o = lfo.crt(lfo.recompose(lfo.icrt(f0_o), wl.B, wl.L))
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    assert (proof_g == proof_o).all() and (lc_g == lc_o).all()
    assert (w0.f == f0_o).all() and (w0.f_coeff == lfo.icrt(f0_o)).all() and (w0.w_ccs == w_ccs_o).all()
    w0.free()


@pytest.mark.parametrize("name", ["D5120", "D10240"])
def test_fold_step_dot_i8_chunk_counts(ctx, name):
    """u_s / eta inner products on the int8 matrix cores at column counts where the number of partial-tile chunks jumps (80 K-steps -> 40
    chunks, 81 -> 27): the scratch is sized for the largest count.  A second context created afterwards checks that nothing was written
    past the scratch into a neighbouring allocation (the proofs of both runs equal the oracle's)."""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name)
    assert wl.n in (5120, 10240)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    for _ in range(2):
        lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
        assert (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()


@pytest.mark.parametrize("name", ["E22", "E99", "E31", "E32"])
def test_fold_step_parity_wider_reference_rows(ctx, name):
    """the reference's wider Goldilocks parameter rows (benches/config.toml:150-165) at small wit_len: kappa 43 / B 2^22 / L 3 / K 22,
    kappa 99 (commit cut into row chunks), and B 2^31 / K 31 (the widest digits the int32 witness planes hold): complete fold steps
    word for word against the oracle, chained once"""
    wl, inst, A, f_coeff, wit, cccs, acc_g, linpr_g, acc_o, linpr_o = run_both(ctx, name)
    assert (acc_g == acc_o).all() and (linpr_g == linpr_o).all()
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    assert (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
    rc, lc_v = inst.verify(lfo.Transcript(), acc_g, cccs, proof_g)
    assert rc == 0 and (lc_v == lc_g).all()
    lc2_g, w2, proof2_g = api.NIFSProver.prove(ctx, lc_g, w0, cccs, wit, api.PoseidonTranscript())
    lc2_o, f2_o, proof2_o = inst.fold_step(lfo.Transcript(), A, lc_o, lfo.icrt(f0_o), cccs, f_coeff)
    assert (proof2_g == proof2_o).all() and (lc2_g == lc2_o).all() and (w2.f == f2_o).all()


def test_b_2_32_edge_digits(ctx):
    """B = 2^32 (benches/config.toml:158): balanced base-B digits lie in [-2^31, 2^31].  The int32 witness planes hold -2^31 (all K = 32
    bit-planes of INT32_MIN must come out right in every kernel that cuts digits: commits, evaluations, look-up codes, compute_f_0) and
    every other value; +2^31 exactly is rejected at ingest (LF_ERR_UNSUPPORTED), never mis-folded."""
    wl = make_workload("E32")
    P = api.P
    h = 1 << 31
    coeff = lfo.icrt(wl.w_ccs)
    coeff[0, 0] = np.uint64(P - h)                       # digit -2^31
    coeff[1, 0] = np.uint64(h - 1)                       # 2^31 - 1
    coeff[2, 1] = np.uint64(P - (h + (h << 31)))         # digits (-2^31, -2^30)
    coeff[3, 5] = np.uint64((h - 1) + ((h >> 1) << 32))  # digits (2^31 - 1, 2^30)
    coeff[4, 7] = np.uint64(P - ((1 << 40) + h))         # digits (-2^31, -2^8)
    wl.w_ccs = lfo.crt(coeff)
    wl.val[2] = np.ascontiguousarray(wl.z()[:min(wl.n, wl.m)])     # C = diag(z) of the modified witness: the CCS stays satisfied
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    assert (f_coeff == P - h).sum() >= 3 and not (f_coeff == h).any()          # -2^31 digits are present, +2^31 is not
    assert (wit.f_coeff == f_coeff).all()
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc, lin = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
    acc_o, lin_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    assert (acc == acc_o).all() and (lin == lin_o).all()
    lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, api.PoseidonTranscript())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    assert (proof == proof_o).all() and (lc == lc_o).all() and (w0.f == f0_o).all()
    rc, _ = inst.verify(lfo.Transcript(), acc, cccs, proof)
    assert rc == 0
    # +2^31: the one digit value the planes cannot hold (here as the tie (2^31 - 2) 2^32 + 2^31, which the balanced rule keeps at +B/2)
    coeff[5, 2] = np.uint64(((h - 2) << 32) + h)
    with pytest.raises(api.LfError) as e:
        api.Witness.from_w_ccs(ctx, lfo.crt(coeff))
    assert e.value.code == -3


@pytest.mark.parametrize("name", ["T10", "G5"])
def test_fold_step_fused_fix_rounds_match_oracle(ctx, name, monkeypatch):
    """rounds >= 4 of the folding sumcheck with fix_variables fused into the round kernel (the driver uses it from 2^14
    table entries on; LF_FOLD_FUSE_MIN lowers the threshold so the oracle-sized cases take that path) and with the
    separate k_fix pass: identical proofs, both equal to the oracle's."""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, 2)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    monkeypatch.setenv("LF_NO_TAIL", "1")          # per-round launches for every round (the persistent tail has its own test below)
    monkeypatch.setenv("LF_FOLD_FUSE_MIN", "4")
    monkeypatch.setenv("LF_FOLD_NO_LUT", "1")
    lc_f, w_f, proof_f = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    # rounds 3 and 4 from the 81-entry digit look-up table instead of materialised m/4-entry tables (large instances by default)
    monkeypatch.delenv("LF_FOLD_NO_LUT")
    monkeypatch.setenv("LF_FOLD_LUT_MIN", "1")
    monkeypatch.setenv("LF_FOLD_TAB_MIN", "1")     # ... and rounds 1-2 as gathers from the per-table coefficient tables
    monkeypatch.setenv("LF_FOLD_TAB_R1", "1")
    lc_l, w_l, proof_l = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    # the product-free forms of rounds 4 and 5 (modes 6 and 7: squares and mu products of the fixed look-up values from tables over the digit codes),
    # with round 5 still on the planes / on the stored round-4 tables / both through the older modes 4 and 1
    # (LF_FOLD_SPLIT_MIN=1: rounds 4 / 5 in the split form -- three sums per slot, the host completes the message -- at this size too)
    for extra in ({"LF_FOLD_R5_MIN": "1"}, {"LF_FOLD_R5_MIN": "1", "LF_FOLD_SPLIT_MIN": "1"}, {"LF_FOLD_NO_R5TAB": "1"}, {"LF_FOLD_NO_R5TAB": "1", "LF_FOLD_SPLIT_MIN": "1"},
                  {"LF_FOLD_NO_R4TAB": "1"}):
        for key, val in extra.items():
            monkeypatch.setenv(key, val)
        lc_x, w_x, proof_x = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
        for key in extra:
            monkeypatch.delenv(key)
        assert (proof_x == proof_o).all() and (lc_x == lc_o).all() and (w_x.f == f0_o).all(), extra
    monkeypatch.delenv("LF_FOLD_LUT_MIN")
    monkeypatch.delenv("LF_FOLD_TAB_MIN")
    monkeypatch.delenv("LF_FOLD_TAB_R1")
    monkeypatch.delenv("LF_FOLD_FUSE_MIN")
    monkeypatch.setenv("LF_FOLD_UNFUSED", "1")
    monkeypatch.setenv("LF_THETA_EVAL", "1")       # theta from evaluate_mles-style sums instead of the last fix of the sumcheck tables
    monkeypatch.setenv("LF_LIN_U_EVAL", "1")       # likewise u of the linearization
    lc_u, w_u, proof_u = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    assert (proof_f == proof_o).all() and (lc_f == lc_o).all() and (w_f.f == f0_o).all()
    assert (proof_l == proof_o).all() and (lc_l == lc_o).all() and (w_l.f == f0_o).all()
    assert (proof_u == proof_o).all() and (lc_u == lc_o).all()


@pytest.mark.parametrize("name", ["T10", "T8", "E32", "E22"])
def test_fold_step_gemm_rounds_match_oracle(ctx, name, monkeypatch):
    """rounds 1..3 of the folding sumcheck as exact int8 GEMMs on the matrix cores (lf_sv_rounds.hip; the driver uses them from 8192
    pairs on, LF_FOLD_SV_MIN lowers the threshold): one, two and three rounds in that form, followed by the look-up-table rounds, the
    fused-fix rounds or plain tables -- identical proofs, all equal to the oracle's"""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, 2)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    monkeypatch.setenv("LF_FOLD_SV_MIN", "64")
    monkeypatch.setenv("LF_DOT_MIN", "64")          # ... and the u_s / eta inner products as int8 GEMMs at this size too
    m = 1 << wl.s
    # (the GEMMs run against one eq value per pair -- two column tiles -- unless LF_FOLD_SV_NO_SPLIT=1; rounds 4 and 5 from the tables over the digit codes leave three
    # sums per slot and the host completes the message unless LF_FOLD_ROUNDS_NO_SPLIT=1: both forms, same words)
    for rounds, extra in ((1, {}), (2, {}), (3, {}), (2, {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4"}), (3, {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4"}),
                          (3, {"LF_NO_TAIL": "1", "LF_FOLD_UNFUSED": "1"}), (3, {"LF_FOLD_SV_NO_SPLIT": "1"}),
                          (3, {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4", "LF_FOLD_R5_MIN": "1", "LF_NO_TAIL": "1", "LF_FOLD_SPLIT_MIN": "1"}),
                          (3, {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4", "LF_FOLD_R5_MIN": "1", "LF_NO_TAIL": "1", "LF_FOLD_ROUNDS_NO_SPLIT": "1"}),
                          (2, {"LF_FOLD_LUT_MIN": "1", "LF_FOLD_FUSE_MIN": "4", "LF_FOLD_NO_R5TAB": "1", "LF_FOLD_ROUNDS_NO_SPLIT": "1"})):
        monkeypatch.setenv("LF_FOLD_SV_ROUNDS", str(rounds))
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        lc, w, proof = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
        for k in extra:
            monkeypatch.delenv(k)
        want = sum(1 << (r - 1) for r in range(1, rounds + 1) if (m >> r) >= 64)
        assert ctx.fold_paths() == want, (rounds, extra, ctx.fold_paths())
        if "LF_FOLD_ROUNDS_NO_SPLIT" in extra or "LF_FOLD_SPLIT_MIN" not in extra:      # (the default threshold is far above these sizes)
            assert ctx.fold_split_rounds() == 0, (extra, ctx.fold_split_rounds())
        elif wl.s >= 5 and wl.N % 4 == 0:                                               # rounds 4 and 5 took the split form (lf_last_fold_split_rounds)
            assert ctx.fold_split_rounds() == 0b11000, (extra, bin(ctx.fold_split_rounds()))
        assert (proof == proof_o).all() and (lc == lc_o).all() and (w.f == f0_o).all(), (rounds, extra)
    monkeypatch.setenv("LF_FOLD_NO_SV", "1")
    lc, w, proof = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    assert ctx.fold_paths() == 0 and (proof == proof_o).all()


@pytest.mark.parametrize("name", ["T10", "G5", "T8"])
def test_fold_step_persistent_tail_matches_oracle(ctx, name, monkeypatch):
    """the persistent tail kernel (k_fold_tail: all remaining folding-sumcheck rounds in one launch, messages and challenges through
    a host-mapped mailbox) taking over at different rounds, with and without the look-up-table rounds before it: identical proofs,
    all equal to the oracle's"""
    wl, inst, A, f_coeff, wit, cccs, acc_g, _, acc_o, _ = run_both(ctx, name, 5)
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    # lut: rounds 3-4 from the digit look-up tables (k_fold_round modes 3/5 and 4/6) and, when the tail starts later than round 5, round 5 from the
    # planes as well (mode 7; LF_FOLD_R5_MIN=1 lifts its size threshold) -- a tail that starts AT round 5 needs the round-4 tables mode 6 then must store
    for tail_n, lut in (("16384", False), ("16", False), ("4", False), ("64", True), ("16", True), ("0", False)):
        monkeypatch.setenv("LF_TAIL_N", tail_n)
        if lut:
            monkeypatch.setenv("LF_FOLD_LUT_MIN", "1")
            monkeypatch.setenv("LF_FOLD_FUSE_MIN", "4")
            monkeypatch.setenv("LF_FOLD_R5_MIN", "1")
        lc, w, proof = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
        if lut:
            monkeypatch.delenv("LF_FOLD_LUT_MIN")
            monkeypatch.delenv("LF_FOLD_FUSE_MIN")
            monkeypatch.delenv("LF_FOLD_R5_MIN")
        assert (proof == proof_o).all() and (lc == lc_o).all() and (w.f == f0_o).all(), (tail_n, lut)
    # the tail rounds' Fiat-Shamir transcript on the DEVICE sponge (no host round trip at all); the host transcript
    # takes the sponge back afterwards (theta / eta absorbs, rho challenges), so any divergence shows in the proof and the folded instance
    monkeypatch.setenv("LF_DEVICE_TRANSCRIPT", "1")
    for tail_n in ("16384", "16"):
        monkeypatch.setenv("LF_TAIL_N", tail_n)
        lc, w, proof = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
        assert (proof == proof_o).all() and (lc == lc_o).all() and (w.f == f0_o).all(), ("device transcript", tail_n)
    monkeypatch.delenv("LF_DEVICE_TRANSCRIPT")
    # twice in a row on the same context (mailbox epochs, self-resetting counters)
    monkeypatch.setenv("LF_TAIL_N", "16384")
    for _ in range(3):
        lc, w, proof = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
        assert (proof == proof_o).all()


# ---- sumcheck through the ABI, split at the transcript -----------------------------------------------
def test_sumcheck_lin_abi_matches_oracle_proof(ctx):
    """drive lf_sumcheck_lin_{begin,round,end} with a Python-side replay of the Fiat-Shamir schedule (App. B) and
    compare every round message with the oracle's LinearizationProof; also the state-machine errors."""
    wl, inst, A, scheme = setup_case(ctx, "T10")
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    cm = lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_coeff))
    cccs = np.concatenate([cm, wl.x_ccs])
    lc_o, pr_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    # replay: beta challenges
    tr = api.PoseidonTranscript()
    label = int.from_bytes(b"beta_s", "big") % P
    tr.absorb_slice(diag(label)[None, :])
    beta = np.stack([tr.get_challenge() for _ in range(wl.s)])
    z = wl.z()
    tables = np.stack([ctx.mat_vec_mul(j, z) for j in range(wl.t)])
    sc = api.MLSumcheckLin(ctx, tables, beta)
    with pytest.raises(api.LfError) as e:
        sc.prove_round(r_prev=np.zeros(3, dtype=np.uint64))      # "first round should be prover first"
    assert e.value.code == -7
    tr.absorb_slice(diag(wl.s)[None, :]); tr.absorb_slice(diag(wl.d + 1)[None, :])
    npts = wl.d + 2
    r = None
    for rnd in range(wl.s):
        msg = sc.prove_round(r)
        assert (msg == pr_o[rnd * npts:(rnd + 1) * npts]).all(), rnd
        tr.absorb_slice(msg)
        r = tr.get_challenge()
        ring_r = np.zeros(RE, dtype=np.uint64)
        for k in range(8):
            ring_r[3 * k:3 * k + 3] = r
        tr.absorb_slice(ring_r[None, :])
        assert (ring_r == lc_o[rnd]).all()
    with pytest.raises(api.LfError):
        sc.prove_round(r)                                          # "Prover is not active"
    sc.end()


def test_parameter_and_state_errors(ctx):
    wl = make_workload("T8")
    bad = make_workload("T8"); bad.b = 4
    with pytest.raises(api.LfError) as e:
        ctx.load_ccs(bad)
    assert e.value.code == -3                                      # LF_ERR_UNSUPPORTED (b != 2)
    bad = make_workload("T8"); bad.wit_len = 100; bad.w_ccs = bad.w_ccs[:100]
    with pytest.raises(api.LfError) as e:
        ctx.load_ccs(bad)                                          # N = 400 > m = 256: CSError::InvalidSizeBounds
    assert e.value.code in (-6, -1)
    ctx.load_ccs(wl)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa + 1, n=wl.N, seed=1)   # kappa mismatch with the loaded CCS
    acc = np.zeros((ctx.lcccs_len, RE), dtype=np.uint64)
    cccs = np.zeros((ctx.cccs_len, RE), dtype=np.uint64)
    with pytest.raises(api.LfError) as e:
        api.NIFSProver.prove(ctx, acc, wit, cccs, wit, api.PoseidonTranscript())
    assert e.value.code == -1
    api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=1)
    nondiag = acc.copy(); nondiag[0, 3] = 5                         # evaluation point that is not a diagonal challenge
    with pytest.raises(api.LfError) as e:
        api.NIFSProver.prove(ctx, nondiag, wit, cccs, wit, api.PoseidonTranscript())
    assert e.value.code == -3


# ---- general CCS shapes (arbitrary SparseMatrix rows, degree-3 CCS of arith/ccs.rs:14-43) -------------------
@pytest.mark.parametrize("name,ccs", [("T8", "deg3"), ("T8", "multi"), ("G5", "deg3"), ("T10", "multi")])
def test_fold_step_parity_general_ccs(ctx, name, ccs):
    wl = make_workload(name, ccs=ccs)
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc_g, linpr_g = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
    acc_o, linpr_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    assert (linpr_g == linpr_o).all() and (acc_g == acc_o).all()
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    assert (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
    rc, lc_v = inst.verify(lfo.Transcript(), acc_g, cccs, proof_g)
    assert rc == 0 and (lc_v == lc_g).all()


@pytest.mark.parametrize("name,ccs", [("T8", "r1cs"), ("G5", "r1cs"), ("T10", "r1cs"), ("T8", "deg3"), ("T10", "multi"), ("G5", "deg3")])
@pytest.mark.parametrize("env", [{"LF_LIN_SPLIT_MIN": "16"}, {"LF_LIN_SPLIT_MIN": "16", "LF_NO_TAIL": "1"}, {"LF_LIN_SPLIT_MIN": "64", "LF_TAIL_N": "64"},
                                 {"LF_LIN_NO_SPLIT": "1"}])
def test_linearization_split_eq_form(ctx, name, ccs, env, monkeypatch):
    """the large linearization rounds sum E_i[p] h(X, p) at d of the d + 2 points and the host completes the message (run_lin_sumcheck): forced
    onto small instances -- leaving the form into the persistent tail, into the plain rounds (LF_NO_TAIL), after one round only -- the
    linearization proof and the LCCCS must stay the oracle's, for R1CS (d = 2), the degree-3 CCS (d = 3: three evaluated points, cubic
    extrapolation) and matrices with two entries per row; LF_LIN_NO_SPLIT=1 is the plain form of every round"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    wl = make_workload(name, ccs=ccs)
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc_g, linpr_g = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
    acc_o, linpr_o = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    bad = np.nonzero((linpr_g != linpr_o).any(axis=1))[0]
    assert bad.size == 0, f"first differing proof elements {bad[:6]} (round = index // {wl.d + 2})"
    assert (acc_g == acc_o).all()
    nsplit = ctx.lin_split_rounds()         # rounds that really ran in the split form
    if "LF_LIN_NO_SPLIT" in env:
        assert nsplit == 0
    elif "LF_NO_TAIL" in env:
        assert nsplit == wl.s - 3           # rounds 1 .. s - 3: the form is left when a round's tables have fewer than LF_LIN_SPLIT_MIN = 16 entries
    else:
        assert 1 <= nsplit < wl.s
    # an instance that does NOT satisfy the constraint system: the claimed sum of round 1 is not zero, the derived values must still be the true ones
    w_bad = wl.w_ccs.copy(); w_bad[1, 0] = (int(w_bad[1, 0]) + 1) % lfo.P
    f_bad = inst.witness_from_w_ccs(w_bad)
    wit_b = api.Witness.from_w_ccs(ctx, w_bad)
    cccs_b = np.concatenate([wit_b.commit(scheme), wl.x_ccs])
    acc_gb, lin_gb = api.LFLinearizationProver.prove(ctx, cccs_b, wit_b, api.PoseidonTranscript())
    acc_ob, lin_ob = inst.linearize(lfo.Transcript(), cccs_b, f_bad)
    assert (lin_gb == lin_ob).all() and (acc_gb == acc_ob).all()


def test_empty_and_degenerate_inputs(ctx):
    """count = 0 / single-element calls through the ABI (the reference's element-wise maps accept empty vectors)"""
    e = np.zeros((0, RE), dtype=np.uint64)
    assert ctx.crt(e).shape == (0, RE) and ctx.icrt(e).shape == (0, RE)
    one = rnd(77, 1, RE)
    assert (ctx.icrt(ctx.crt(one)) == one).all()
    d = ctx.decompose(one, 1 << 16, 4, 1)
    assert (ctx.recompose(ctx.decompose(one, 1 << 16, 4, 0), 1 << 16, 4) == one).all() and d.shape == (4, RE)
    ok, mx = ctx.linf_check(lfo.crt(np.zeros((1, RE), dtype=np.uint64)), 1)
    assert ok and mx == 0
    A = rnd(5, 2, 1, RE)
    s = api.AjtaiCommitmentScheme(ctx, matrix=A)           # a single column
    f = rnd(6, 1, RE)
    assert (s.commit_ntt(f) == lfo.ajtai_commit(A, 2, 1, f)).all()
    pt = rnd(8, 1, 3)
    assert (ctx.evaluate_mles(rnd(9, 2, 1, RE), pt)[0] == lfo.mle_eval(rnd(9, 2, 1, RE)[0], np.tile(pt, (1, 8)))).all()


def test_fold_real_circuit_from_r1cs(ctx):
    """the reference's own test circuit x^3 + x + 5 = y (arith/r1cs.rs:128-151) through CCS::from_r1cs_padded (latticefold_amd/ccs.py)"""
    from latticefold_amd import ccs
    A3, B3, C3 = ccs.vitalik_r1cs()
    z = ccs.vitalik_z_ntt("goldilocks")
    wl = ccs.workload_from_r1cs(A3, B3, C3, 1, z[0:1], z[2:], ring="goldilocks", L=4, Bbase=1 << 16, K=16, kappa=4, name="vitalik")
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    assert (wit.w_ccs == wl.w_ccs).all()
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc_g, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
    acc_o, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    assert (acc_g == acc_o).all() and (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
    ok, lc_p, _ = api.NIFSVerifier.verify(wl, acc_g, cccs, proof_g, api.PoseidonTranscript())
    assert ok and (lc_p == lc_g).all()


def test_fold_step_with_another_nonresidue(ctx, monkeypatch):
    """nu is data too: F_{p^3} = F_p[Y]/(Y^3 - w^5) (w = 2^40) switches every kernel to its generic-nu instantiation
    (no 2^40 shift tricks); CRT and a whole fold step must still match the oracle under the same tables."""
    nr, y = ctx.get_ring_tables()
    w = 1 << 40
    nu2 = pow(w, 5, P)
    E = [1, 5, 7, 11, 13, 17, 19, 23]
    y2 = np.zeros((8, 3), dtype=np.uint64)
    for k, e in enumerate(E):
        g = 1 if e % 3 == 2 else 2                   # (c Y^g)^3 = c^3 nu^g = w^e  needs  3 | e - 5 g
        a = ((e - 5 * g) % 24) // 3
        assert (e - 5 * g) % 3 == 0
        y2[k, g] = pow(w, a, P)
    try:
        ctx.set_ring_tables(nu2, y2.reshape(-1))
        assert lfo.lib().lfo_set_ring(nu2, lfo._p64(np.ascontiguousarray(y2.reshape(-1)))) == 0
        x = rnd(5, 200, RE)
        assert (ctx.crt(x) == lfo.crt(x)).all() and (ctx.icrt(ctx.crt(x)) == x).all()
        assert ctx.selftest_field(7, 1 << 16) == 0
        wl, inst, A, f_coeff, wit, cccs, acc_g, linpr_g, acc_o, linpr_o = run_both(ctx, "T8")
        assert (linpr_g == linpr_o).all() and (acc_g == acc_o).all()
        lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
        lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
        assert (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
        rc, _ = inst.verify(lfo.Transcript(), acc_g, cccs, proof_g)
        assert rc == 0
        # the look-up-table / fused forms of rounds 2-6 in their generic-nu instantiations
        for k in ("LF_FOLD_LUT_MIN", "LF_FOLD_TAB_MIN", "LF_FOLD_TAB_R1"):
            monkeypatch.setenv(k, "1")
        monkeypatch.setenv("LF_FOLD_FUSE_MIN", "4")
        lc_l, w_l, proof_l = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
        assert (proof_l == proof_o).all() and (lc_l == lc_o).all() and (w_l.f == f0_o).all()
    finally:
        for k in ("LF_FOLD_LUT_MIN", "LF_FOLD_TAB_MIN", "LF_FOLD_TAB_R1", "LF_FOLD_FUSE_MIN"):
            monkeypatch.delenv(k, raising=False)
        ctx.set_ring_tables(nr, y)
        lfo.lib().lfo_set_ring(nr, lfo._p64(np.ascontiguousarray(y)))


def test_fold_step_with_several_public_inputs(ctx):
    """x_len = 3 (all bench workloads use 1): x_s decomposition, z heads, x_0 folding with l + 1 = 4 head elements"""
    wl = make_workload("T8", l=3)
    inst = lfo.Instance(wl)
    ctx.load_ccs(wl)
    A = wl.ajtai_matrix()
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
    cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
    acc_g, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
    acc_o, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    lc_g, w0, proof_g = api.NIFSProver.prove(ctx, acc_g, wit, cccs, wit, api.PoseidonTranscript())
    lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f_coeff, cccs, f_coeff)
    assert (acc_g == acc_o).all() and (proof_g == proof_o).all() and (lc_g == lc_o).all() and (w0.f == f0_o).all()
    ok, lc_p, _ = api.NIFSVerifier.verify(wl, acc_g, cccs, proof_g, api.PoseidonTranscript())
    assert ok and (lc_p == lc_g).all()
