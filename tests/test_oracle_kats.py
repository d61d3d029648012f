"""Pin the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md 4-B / 8c).  Vectors: tests/golden/kats.json (extracted by tests/tools/extract_kats.py)."""
import numpy as np
import pytest

import lfo
from latticefold_amd.workload import P, RE, diag, make_workload

rng = np.random.default_rng(1234)


def rand_fq(*shape):
    return (rng.integers(0, 2**63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)) % np.uint64(P)


def test_field_reduction_matches_bigint():
    a = rand_fq(2000); b = rand_fq(2000)
    # fq3 product with c1=c2=0 exercises fq_mul's fast reduction
    for x, y in zip(a[:200], b[:200]):
        o = np.zeros(3, dtype=np.uint64)
        lfo.lib().lfo_fq3_mul(lfo._p64(np.array([x, 0, 0], dtype=np.uint64)), lfo._p64(np.array([y, 0, 0], dtype=np.uint64)), lfo._p64(o))
        assert int(o[0]) == int(x) * int(y) % P and o[1] == 0 and o[2] == 0
    for x, y in [(P - 1, P - 1), (P - 1, 1), (2**32, 2**32), (2**63, 2), (0xFFFFFFFF, 0xFFFFFFFF00000000)]:
        o = np.zeros(3, dtype=np.uint64)
        lfo.lib().lfo_fq3_mul(lfo._p64(np.array([x, 0, 0], dtype=np.uint64)), lfo._p64(np.array([y, 0, 0], dtype=np.uint64)), lfo._p64(o))
        assert int(o[0]) == x * y % P


def test_poseidon_params_match_reference_table(kats):
    k = kats["poseidon_goldilocks_params"]
    ark = np.zeros(720, dtype=np.uint64); mds = np.zeros(576, dtype=np.uint64)
    lfo.lib().lfo_poseidon_params(lfo._p64(ark), lfo._p64(mds))
    assert [int(x) for x in ark[:4]] == k["ark_first"] and [int(x) for x in ark[-4:]] == k["ark_last"]
    assert [int(x) for x in mds[:4]] == k["mds_first"] and [int(x) for x in mds[-4:]] == k["mds_last"]
    assert sum((i + 1) * int(v) for i, v in enumerate(ark)) % P == k["ark_checksum"]
    assert sum((i + 1) * int(v) for i, v in enumerate(mds)) % P == k["mds_checksum"]


def test_poseidon_big_challenge(kats):
    k = kats["poseidon_big_challenge"]
    tr = lfo.Transcript()
    tr.absorb_fq(k["absorb"])
    assert [int(x) for x in tr.challenge()] == k["expected_fq3"]


def test_poseidon_small_challenge(kats):
    k = kats["poseidon_small_challenge"]
    tr = lfo.Transcript()
    tr.absorb_fq(k["absorb"])
    assert [int(x) for x in tr.short_challenge()] == k["expected_coeffs"]


def test_small_challenge_from_bytes(kats):
    k = kats["goldilocks_small_challenge_from_bytes"]
    assert [int(x) for x in lfo.short_challenge_from_bytes(k["bytes"])] == k["coeffs"]


def test_rot_lin_combination(kats):
    k = kats["rot_lin_combination"]
    rho = np.array(k["rho_coeff"], dtype=np.uint64)
    theta = np.array(k["theta"], dtype=np.uint64)
    out = lfo.rot_lin_combination(rho, theta, 3)
    assert out.tolist() == k["expected"]


def test_rot_sum_is_ring_product():
    """test_rot_sum_with_coeffs (rotation.rs:115-136): RotSum(a, coeff(b)) = coeff(a*b)."""
    a = rand_fq(RE); b = rand_fq(RE)
    theta = np.zeros((3, RE), dtype=np.uint64)   # 24 Fq3 entries = b embedded
    theta.reshape(-1)[0::3] = b
    got = lfo.rot_lin_combination(a[None, :], theta[None, :, :], 1).reshape(-1)
    want = np.zeros(RE, dtype=np.uint64)
    lfo.lib().lfo_ring_mul_coeff(lfo._p64(a), lfo._p64(b), lfo._p64(want))
    assert got[0::3].tolist() == want.tolist() and not got[1::3].any() and not got[2::3].any()


def test_crt_is_ring_isomorphism():
    a = rand_fq(16, RE); b = rand_fq(16, RE)
    assert (lfo.icrt(lfo.crt(a)) == a).all()
    prod = np.zeros_like(a)
    for i in range(16):
        lfo.lib().lfo_ring_mul_coeff(lfo._p64(a[i]), lfo._p64(b[i]), lfo._p64(prod[i]))
    ntt_prod = np.zeros_like(a)
    lfo.lib().lfo_ring_mul_ntt(lfo._p64(lfo.crt(a)), lfo._p64(lfo.crt(b)), lfo._p64(ntt_prod), 16)
    assert (lfo.crt(prod) == ntt_prod).all()
    # constants embed diagonally (R::from(u128)), consistent with test_commit_ntt
    c = np.zeros((1, RE), dtype=np.uint64); c[0, 0] = 12345
    assert (lfo.crt(c)[0] == diag(12345)).all()


def test_commit_ntt_closed_form(kats):
    k = kats["commit_ntt"]
    kappa, n = 3, 1 << 10   # same closed form at a size the oracle finishes instantly
    A = np.stack([np.stack([diag(i * n + j) for j in range(n)]) for i in range(kappa)])
    f = np.tile(diag(2), (n, 1))
    out = lfo.ajtai_commit(A, kappa, n, f)
    for i in range(kappa):
        assert (out[i] == diag(n * (2 * i * n + (n - 1)))).all()


def test_get_fhat_layout(kats):
    """f-hat is virtual in both implementations; its layout is pinned through the evaluation v = fhat_d(r):
    with r a hypercube point (index i) the evaluation returns entry i of table d."""
    k = kats["get_fhat"]
    wl = make_workload("T8")
    inst = lfo.Instance(wl)
    f = np.zeros((wl.N, RE), dtype=np.uint64)
    f[0] = k["f_coeff"][0]; f[1] = k["f_coeff"][1]
    for i in (0, 1):
        r = np.stack([diag((i >> j) & 1) for j in range(wl.s)])
        for d in range(3):
            tab = np.zeros((wl.m, RE), dtype=np.uint64)
            tab[:, 0::3] = f[:, 8 * d:8 * d + 8]
            got = lfo.mle_eval(tab, r)
            assert got[0::3].tolist() == k["fhat_slots"][d][i]


@pytest.mark.parametrize("mode", [0])  # mode 1 (digits in [-b/2,b/2)) cannot represent (p-1)/2 with B^L = 2^64
def test_decompose_recompose_roundtrip(mode):
    """nifs/decomposition/utils.rs:84-121 (Style-A): sum_k b^k digit_k == value, digits small."""
    lfo.lib().lfo_set_digit_mode(mode)
    try:
        x = rand_fq(8, RE)
        x[0, :6] = [0, 1, P - 1, (P - 1) // 2, (P + 1) // 2, 2**15]
        for base, digits in ((1 << 16, 4), (1 << 15, 5)):
            d = lfo.decompose(x, base, digits, 0)
            assert (lfo.recompose(d, base, digits) == x).all()
            dv = d.astype(object); signed = np.where(dv > P // 2, dv - P, dv)
            assert max(abs(int(v)) for v in signed.reshape(-1)) <= base // 2
        small = lfo.decompose(x, 1 << 16, 4, 0)
        parts = lfo.decompose(small, 2, 16, 1).reshape(16, -1, RE)
        acc = np.zeros_like(small, dtype=object)
        for kk in range(16):
            pv = parts[kk].astype(object); acc += np.where(pv > P // 2, pv - P, pv) * (1 << kk)
            assert set(np.unique(parts[kk])) <= {0, 1, P - 1}
        assert ((acc % P) == small.astype(object)).all()
    finally:
        lfo.lib().lfo_set_digit_mode(0)
