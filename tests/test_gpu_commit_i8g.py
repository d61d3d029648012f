"""General Ajtai commitments on the int8 matrix cores (lf_ajtai_i8g.hip) against the oracle: AjtaiCommitmentScheme::commit_ntt
(commitment/commitment_scheme.rs:37-54,75-77) on arbitrary vectors -- edge residues (0, 1, p-1, (p +- 1)/2), the digit patterns that drive
the int8 operands and the int32 accumulators to their limits (all base-128 digits -64 / 63 against all-0x00 / all-0xFF bytes of A, a column
chunk longer than the accumulator flush period), ragged widths, row chunks -- and Witness::commit (arith.rs:357-362) from the int32 planes
of a witness handle."""
import os

import numpy as np
import pytest

from latticefold_amd import api
from latticefold_amd.workload import make_workload, splitmix_fq

pytestmark = pytest.mark.gpu

P_G = 0xFFFFFFFF00000001
P_B = 15 * 2**27 + 1


def _oracle(ring):
    if ring == "goldilocks":
        import lfo as O
    else:
        import lfo_bb as O
    return O


def _ctx(ring, env=None):
    for k in ("LF_I8G_WGS",):
        os.environ.pop(k, None)
    for k, v in (env or {}).items():
        os.environ[k] = v
    return api.Context(0, ring=ring)


def _rnd(seed, p, *shape):
    n = int(np.prod(shape))
    return (splitmix_fq(seed, 0, n) % np.uint64(p)).reshape(shape)


@pytest.mark.parametrize("ring", ["goldilocks", "babybear"])
@pytest.mark.parametrize("kappa,n", [(3, 1), (5, 7), (9, 64), (16, 333), (26, 1000), (20, 777), (13, 8 * 41), (32, 130)])
def test_edge_residues(ring, kappa, n):
    O = _oracle(ring)
    p = P_G if ring == "goldilocks" else P_B
    ctx = _ctx(ring)
    try:
        if ring == "babybear" and kappa > 32:
            pytest.skip("backend limit")
        RE = ctx.RE
        A = _rnd(100 + kappa, p, kappa, n, RE)
        edge = np.array([0, 1, p - 1, (p - 1) // 2, (p + 1) // 2, 2, p - 2, 127, 128, p - 128, p - 129, 1 << 31, (1 << 31) - 1], dtype=np.uint64) % np.uint64(p)
        f_coeff = edge[(np.arange(n * RE) * 7 + np.arange(n * RE) // 5) % len(edge)].reshape(n, RE)
        for f in (O.crt(f_coeff), f_coeff, _rnd(7, p, n, RE)):     # edge coefficients, edge slots, random
            f = np.ascontiguousarray(f)
            got = api.AjtaiCommitmentScheme(ctx, matrix=A).commit_ntt(f)
            assert (got == O.ajtai_commit(A, kappa, n, f)).all()
    finally:
        ctx.close()


def _digits_to_value(dig, p):
    """coefficient whose balanced base-128 digits are all `dig` (9 planes for the 64-bit ring, 4 for the 31-bit one: inside the centred range)"""
    k = 9 if p == P_G else 4
    v = dig * (128**k - 1) // 127
    assert abs(v) <= (p - 1) // 2
    return v % p


@pytest.mark.parametrize("ring", ["goldilocks", "babybear"])
@pytest.mark.parametrize("dig,abyte", [(-64, 0x00), (-64, 0xFF), (63, 0x00), (63, 0xFF)])
def test_extreme_operands_and_accumulator_flush(ring, dig, abyte):
    """every Toeplitz entry at its int8 limit (-128 / 126) against every byte of A at its limit, over a column chunk longer than the flush period
    (two workgroups per row half: 2 x 700 tiles > 682 (Goldilocks) / 227 (BabyBear) tiles)"""
    O = _oracle(ring)
    p = P_G if ring == "goldilocks" else P_B
    ctx = _ctx(ring, {"LF_I8G_WGS": "4" if ring == "goldilocks" else "2"})
    try:
        RE, kappa, n = ctx.RE, 26 if ring == "goldilocks" else 16, 8 * 1400 - 3
        nb = 8 if ring == "goldilocks" else 4
        a = int.from_bytes(bytes([abyte]) * nb, "little")
        if a >= p:
            a = (0xFFFFFFFF00000000 if ring == "goldilocks" else p - 1)      # the largest canonical residue: upper bytes 0xFF
        A_coeff = np.full((kappa, n, RE), a, dtype=np.uint64)
        f_coeff = np.full((n, RE), _digits_to_value(dig, p), dtype=np.uint64)
        A = np.ascontiguousarray(O.crt(A_coeff.reshape(kappa * n, RE)).reshape(kappa, n, RE))
        f = np.ascontiguousarray(O.crt(f_coeff))
        got = api.AjtaiCommitmentScheme(ctx, matrix=A).commit_ntt(f)
        assert (got == O.ajtai_commit(A, kappa, n, f)).all()
    finally:
        os.environ.pop("LF_I8G_WGS", None)
        ctx.close()


@pytest.mark.parametrize("name", ["T8", "T10", "E99", "E32", "G5", "D5120", "B6", "B10", "B21", "B333"])
def test_witness_commit_from_int32_planes(name):
    wl = make_workload(name)
    O = _oracle(wl.ring)
    ctx = _ctx(wl.ring)
    try:
        ctx.load_ccs(wl)
        A = wl.ajtai_matrix()
        scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
        inst = O.Instance(wl)
        f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
        w = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        want = O.ajtai_commit(A, wl.kappa, wl.N, O.crt(f_coeff))
        assert (w.commit(scheme) == want).all()
        # coefficients at the bound: +-(B/2 - 1) and +-B/2 where the handle accepts them
        p = P_G if wl.ring == "goldilocks" else P_B
        half = wl.B // 2
        pat = np.array([half - 1, p - (half - 1), 0, 1, p - 1, half - 2], dtype=np.uint64)
        fc = pat[(np.arange(wl.N * ctx.RE) * 5) % len(pat)].reshape(wl.N, ctx.RE)
        w2 = api.Witness.from_f_coeff(ctx, fc)
        assert (w2.commit(scheme) == O.ajtai_commit(A, wl.kappa, wl.N, O.crt(fc))).all()
    finally:
        ctx.close()
