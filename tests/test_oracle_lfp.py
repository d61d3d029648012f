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
