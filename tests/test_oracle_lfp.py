"""LatticeFold+ oracle slice (oracle/lfp.c) on the CPU: the reference's tensor KATs (crates/latticefold-plus/src/utils.rs:118-131), ring
arithmetic identities of Z_p[X]/(X^16 + 1), and the internal consistency of RgInstance::from_f (rgchk.rs:260-331): the k monomial
matrices recompose the witness, tau recomposes the double commitment, exp / shifts agree with ring multiplication."""
import json
import os

import numpy as np

import lfp

KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))["lfp_tensor"]
P, D = lfp.P, lfp.D


def test_tensor_kats():
    k = KATS["tensor_product"]
    assert lfp.tensor_product(k["a"], k["b"]).tolist() == [v % P for v in k["expected"]]
    k = KATS["tensor"]
    assert lfp.tensor(k["r"]).tolist() == [v % P for v in k["expected"]]
    assert lfp.tensor_product([], [3, 4]).tolist()[:2] == [3, 4]      # an empty side returns the other (utils.rs:52-57)


def test_ring_is_negacyclic():
    x = np.zeros(D, dtype=np.uint64); x[1] = 1
    a = lfp.splitmix(1, 0, D)
    r = a.copy()
    for _ in range(D):
        r = lfp.ring_mul(r, x)
    assert (r == (np.uint64(P) - a) % np.uint64(P)).all()             # X^16 = -1
    b, c = lfp.splitmix(2, 0, D), lfp.splitmix(3, 0, D)
    assert (lfp.ring_mul(lfp.ring_mul(a, b), c) == lfp.ring_mul(a, lfp.ring_mul(b, c))).all()


def test_rg_from_f_is_consistent():
    n, kappa, b, k = 1 << 14, 1, 8, 2
    l = int(np.ceil(np.log(P) / np.log(b)))
    assert l == 22
    A = lfp.splitmix(7, 0, kappa * n * D).reshape(kappa, n, D)
    small = (lfp.splitmix(8, 0, n * D) % np.uint64(63)).astype(np.int64) - 31      # |coefficient| <= 31 < b^k / 2
    f = np.array([int(v) % P for v in small], dtype=np.uint64).reshape(n, D)
    r = lfp.rg_from_f(f, A, b, k, l)
    # digits recompose the coefficients and stay inside the exp domain (-d/2, d/2)
    rec = sum(r["Df"][ki].astype(np.int64) * b ** ki for ki in range(k))
    assert (rec == small.reshape(n, D)).all() and np.abs(r["Df"]).max() < D // 2
    # comM_f against explicit ring products with the monomials exp(D_f)
    part = lfp.rg_from_f(np.concatenate([f[:64], np.zeros((n - 64, D), dtype=np.uint64)]), A, b, k, l)      # same first 64 rows, rest zero
    rest = np.zeros(D, dtype=object)                                                                         # exp(0) = 1 for the zero rows
    for j in range(64, n):
        rest = (rest + A[0, j].astype(object)) % P
    for ki in range(k):
        for c in (0, 5, 15):
            acc = np.zeros(D, dtype=np.uint64)
            for j in range(64):
                m = np.zeros(D, dtype=np.uint64)
                e = int(r["Df"][ki, j, c]); m[e if e >= 0 else D + e] = 1
                acc = (acc.astype(object) + lfp.ring_mul(A[0, j], m).astype(object)) % P
            assert ((acc + rest) % P == part["comMf"][ki, 0, c].astype(object)).all()
    # tau holds base-(d/2) digits that recompose the double commitment, zero padded
    need = kappa * k * D * l * D
    assert not r["tau"][need:].any()
    tc = np.where(r["tau"] > P // 2, r["tau"].astype(object) - P, r["tau"].astype(object))
    assert max(abs(int(v)) for v in tc[:need]) <= D // 4
    pos = 0
    for i in range(kappa):
        for ki in range(k):
            for c in range(D):
                chunk = np.array(tc[pos:pos + l * D], dtype=object).reshape(l, D)
                val = sum(chunk[j] * (D // 2) ** j for j in range(l)) % P
                assert (val == r["comMf"][ki, i, c].astype(object)).all()
                pos += l * D
    # the three commitments
    assert (r["cm_f"] == lfp.commit(A, f)).all()
    tau_ring = np.zeros((n, D), dtype=np.uint64); tau_ring[:, 0] = r["tau"]
    assert (r["C_Mf"] == lfp.commit(A, tau_ring)).all()
    m_tau = np.zeros((n, D), dtype=np.uint64)
    for j in range(n):
        e = int(tc[j]); m_tau[j, e if e >= 0 else D + e] = 1
    assert (r["cm_mtau"] == lfp.commit(A, m_tau)).all()


def _sparse(seed, n, per_row, scale_first=True):
    """n x n CSR with ring-element coefficients: identity plus (per_row - 1) random off-diagonal ring elements per row; entry (0, 0) doubled as in the
    reference's tests (decomp.rs:167-169)"""
    rowptr = np.arange(n + 1, dtype=np.uint32) * per_row
    col = np.zeros(n * per_row, dtype=np.uint32)
    val = np.zeros((n * per_row, D), dtype=np.uint64)
    rnd = lfp.splitmix(seed, 0, n * per_row * (D + 1))
    for r in range(n):
        col[r * per_row] = r
        val[r * per_row, 0] = 1
        for k in range(1, per_row):
            i = (r * per_row + k) * (D + 1)
            col[r * per_row + k] = int(rnd[i]) % n
            val[r * per_row + k] = rnd[i + 1:i + 1 + D]
    if scale_first:
        val[0, 0] = 2
    return rowptr, col, val


def test_decompose_recomposes():
    """DecompProof::verify (decomp.rs:101-123): recompose([C0, C1], B) = A f and recompose([v0, v1], B) = the evaluations of f (and of M_j f)"""
    n, kappa, B = 256, 2, 50
    A = lfp.splitmix(21, 0, kappa * n * D).reshape(kappa, n, D)
    small = (lfp.splitmix(22, 0, n * D) % np.uint64(2 * 1200 + 1)).astype(np.int64) - 1200          # |coefficient| <= 1200 < B^2 / 2
    f = np.array([int(v) % P for v in small], dtype=np.uint64).reshape(n, D)
    r_a, r_b = lfp.splitmix(23, 0, 8 * D).reshape(8, D), lfp.splitmix(24, 0, 8 * D).reshape(8, D)
    mats = [_sparse(31, n, 1), _sparse(32, n, 3)]
    d = lfp.decompose(f, A, B, r_a, r_b, mats)
    # digits: f = F0 + B F1 with |digit| <= B / 2
    c = lambda x: np.where(x > P // 2, x.astype(object) - P, x.astype(object))
    assert (c(d["F0"]) + B * c(d["F1"]) == small.reshape(n, D)).all() and max(abs(int(v)) for v in c(d["F0"]).reshape(-1)) <= B // 2
    rec = lambda x0, x1: (x0.astype(object) + B * x1.astype(object)) % P
    assert (rec(d["C0"], d["C1"]) == lfp.commit(A, f).astype(object)).all()
    # the same decomposition of the "whole" vector: evaluations are linear, so decompose(f) with B large enough that F1 = 0 gives them directly
    whole = lfp.decompose(f, A, 1 << 40, r_a, r_b, mats)
    assert not whole["F1"].any() and (whole["F0"] == f).all()
    assert (rec(d["v0"], d["v1"]) == whole["v0"].astype(object)).all()
    # an evaluation by hand: at a Boolean point the MLE returns the table entry (variable 0 = index bit 0)
    idx = 0b10110101
    pt = np.zeros((8, D), dtype=np.uint64)
    for k in range(8):
        pt[k, 0] = (idx >> k) & 1
    e = lfp.decompose(f, A, 1 << 40, pt, pt, [])
    assert (e["v0"][0, 0] == f[idx]).all()
