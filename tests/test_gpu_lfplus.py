"""LatticeFold+ double commitment on the GPU (include/lfplus.h) against the oracle (oracle/lfp.c), bit for bit: RgInstance::from_f at the
sizes of the reference's tests (rgchk.rs:353-420: n = 2^15, kappa = 1, k = 2) and benches (double_commitment: (32768, k 2, kappa 2),
(65536, k 4, kappa 2)), ragged n, wide kappa / k groups, a non-power-of-two digit base, the error paths the reference panics on, and the
tensor KATs of utils.rs:118-131 on the device."""
import json
import os

import numpy as np
import pytest

import lfp as O
from latticefold_amd import plus

pytestmark = pytest.mark.gpu
P, D = plus.P, plus.D
KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))["lfp_tensor"]


def small_f(seed, n, bound):
    """ring elements with centred coefficients in [-bound, bound]"""
    v = (O.splitmix(seed, 0, n * D) % np.uint64(2 * bound + 1)).astype(np.int64) - bound
    return np.where(v < 0, np.uint64(P) - (-v).astype(np.uint64), v.astype(np.uint64)).reshape(n, D)


@pytest.fixture(scope="module")
def ctx():
    c = plus.PlusContext(0)
    yield c
    c.close()


def check_from_f(ctx, n, kappa, k, b=8, seed=1, f=None):
    dp = plus.DecompParameters.for_frog(k, b)
    bound = b ** k // 2 - 1
    A = O.splitmix(seed, 0, kappa * n * D).reshape(kappa, n, D)
    if f is None:
        f = small_f(seed + 1, n, bound)
    want = O.rg_from_f(f, A, dp.b, dp.k, dp.l)
    got = plus.RgInstance.from_f(ctx, f, A, dp)
    assert (got.D_f == want["Df"]).all()
    assert (got.comM_f == want["comMf"]).all()
    assert (got.tau == want["tau"]).all()
    tc = np.where(want["tau"] > P // 2, -(np.uint64(P) - want["tau"]).astype(np.int64), want["tau"].astype(np.int64))
    assert (got.m_tau_exp == tc).all()
    assert (got.fcoms.cm_f == want["cm_f"]).all()
    assert (got.fcoms.C_Mf == want["C_Mf"]).all()
    assert (got.fcoms.cm_mtau == want["cm_mtau"]).all()
    return got, A


def test_reference_test_shape(ctx):
    """rgchk.rs:353-375: f = [2 + 5X, 4 + X^2, 0, ...], n = 2^15, kappa = 1, k = 2, b = d/2"""
    n = 1 << 15
    f = np.zeros((n, D), dtype=np.uint64)
    f[0, 0], f[0, 1], f[1, 0], f[1, 2] = 2, 5, 4, 1
    got, A = check_from_f(ctx, n, 1, 2, f=f)
    # D_f of the two non-zero rows by hand: 5 = -3 + 1*8, 4 = 4 + 0*8
    assert got.D_f[0, 0, :2].tolist() == [2, -3] and got.D_f[1, 0, :2].tolist() == [0, 1] and got.D_f[0, 1, 0] == 4
    assert (plus.exp(got.D_f[0, 0])[1] == np.eye(D, dtype=np.uint64)[D - 3]).all()


@pytest.mark.parametrize("n,kappa,k", [(32768, 2, 2), (65536, 2, 4), (1 << 15, 1, 2), (40000, 1, 3), (33000, 2, 2), (1 << 17, 5, 1), (1 << 18, 1, 6)])
def test_from_f_matches_the_oracle(ctx, n, kappa, k):
    check_from_f(ctx, n, kappa, k, seed=n % 97 + kappa)


def test_general_digit_base(ctx):
    check_from_f(ctx, 1 << 15, 1, 2, b=6, seed=5)          # not a power of two: the division path
    check_from_f(ctx, 1 << 15, 1, 3, b=4, seed=6)


def test_commit_general_vector(ctx):
    n, kappa = 5000, 3
    A = O.splitmix(11, 0, kappa * n * D).reshape(kappa, n, D)
    v = O.splitmix(12, 0, n * D).reshape(n, D)                # full-size coefficients
    ctx.set_matrix(A)
    assert (ctx.commit(v) == O.commit(A, v)).all()
    v[:] = P - 1
    assert (ctx.commit(v) == O.commit(A, v)).all()


def test_error_paths(ctx):
    n = 1 << 15
    A = O.splitmix(3, 0, n * D).reshape(1, n, D)
    dp = plus.DecompParameters.for_frog(2)
    f = small_f(4, n, 31)
    f[77, 3] = 1000                                           # needs more than k base-8 digits: k digits are taken, the rest is dropped
    inst = plus.RgInstance.from_f(ctx, f, A, dp)
    assert (inst.D_f == O.rg_from_f(f, A, dp.b, dp.k, dp.l)["Df"]).all()
    with pytest.raises(plus.LfPlusError) as e:               # base 16: a digit can be +-8, outside the exp domain
        f2 = f.copy()
        f2[5, 5] = 8
        plus.RgInstance.from_f(ctx, f2, A, plus.DecompParameters(16, 2, dp.l))
    assert e.value.code == plus.E_EXP_DOMAIN
    with pytest.raises(plus.LfPlusError) as e:               # split: tau does not fit below n
        plus.RgInstance.from_f(ctx, f[:8192], A[:, :8192], dp)
    assert e.value.code == plus.E_SMALL_N
    with pytest.raises(plus.LfPlusError) as e:
        bad = f.copy()
        bad[0, 0] = P
        plus.RgInstance.from_f(ctx, bad, A, dp)
    assert e.value.code == plus.E_ARG


def test_tensor_kats_on_the_device(ctx):
    k = KATS["tensor_product"]
    assert ctx.tensor_product(k["a"], k["b"]).tolist() == [v % P for v in k["expected"]]
    k = KATS["tensor"]
    assert ctx.tensor(k["r"]).tolist() == [v % P for v in k["expected"]]
    assert ctx.tensor_product([], [3, 4]).tolist() == [3, 4]
    r = O.splitmix(9, 0, 12)
    assert (ctx.tensor(r) == O.tensor(r)).all()
    a, b = O.splitmix(10, 0, 37), O.splitmix(11, 0, 53)
    assert (ctx.tensor_product(a, b) == O.tensor_product(a, b)).all()


def _sparse(seed, n, per_row):
    rowptr = np.arange(n + 1, dtype=np.uint32) * per_row
    col = np.zeros(n * per_row, dtype=np.uint32)
    val = np.zeros((n * per_row, D), dtype=np.uint64)
    rnd = O.splitmix(seed, 0, n * per_row * (D + 1)).reshape(n * per_row, D + 1)
    col[:] = (rnd[:, 0] % np.uint64(n)).astype(np.uint32)
    val[:] = rnd[:, 1:]
    col[::per_row] = np.arange(n, dtype=np.uint32)          # first entry of a row: the diagonal, coefficient 1 (2 in row 0: decomp.rs:167-169)
    val[::per_row] = 0
    val[::per_row, 0] = 1
    val[0, 0] = 2
    return rowptr, col, val


@pytest.mark.parametrize("n,kappa,B,nm", [(1 << 15, 2, 50, 3), (1 << 12, 1, 3989010971, 1), (1 << 10, 3, 7, 0)])
def test_decompose_matches_the_oracle(ctx, n, kappa, B, nm):
    """Decomp::decompose (decomp.rs:32-99) at the reference's test shape (n = 2^15, kappa 2, B = 50: decomp.rs:159-163; B = ceil(sqrt q) + 1:
    decomp.rs:197-200), word for word against the oracle, and the identities DecompProof::verify checks"""
    A = O.splitmix(41, 0, kappa * n * D).reshape(kappa, n, D)
    bound = min(B * B // 2 - 1, P // 2)
    f = small_f(42, n, min(bound, 1 << 40)) if bound < (1 << 62) else O.splitmix(42, 0, n * D).reshape(n, D)
    nv = n.bit_length() - 1
    r = O.splitmix(43, 0, nv * 2 * D).reshape(nv, 2, D)
    mats = [_sparse(50 + j, n, 1 + j) for j in range(nm)]
    got = ctx.decompose(f, A, B, r, mats)
    want = O.decompose(f, A, B, r[:, 0], r[:, 1], mats)
    for k in ("F0", "F1", "C0", "C1", "v0", "v1"):
        assert (got[k] == want[k]).all(), k
    rec = lambda x0, x1: (x0.astype(object) + B * x1.astype(object)) % P
    assert (rec(got["C0"], got["C1"]) == ctx.commit(f).astype(object)).all()
