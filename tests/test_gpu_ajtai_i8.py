"""The int8 matrix-core commit kernels (lf_ajtai_i8.hip) against the oracle: the K-1 digit-plane commitments of LFDecompositionProver::prove
(decomposition.rs:178-201), word for word, over shapes that exercise ragged column tiles (n not a multiple of 8 * workgroups), odd kappa (padded row
tile), row chunks (kappa > 26), plane groups (K - 1 > 16), extreme digits (all +1 / all -1 / alternating: the int32 accumulators and the -128 byte
bias), and a few workgroup counts (with and without the L2 coupling of the paired plane-group workgroups)."""
import os

import numpy as np
import pytest

from latticefold_amd import api
from latticefold_amd.workload import make_workload

pytestmark = pytest.mark.gpu


def _setup(name, env=None):
    for k in ("LF_I8_WGS", "LF_I8_COUPLE_W"):
        os.environ.pop(k, None)
    for k, v in (env or {}).items():
        os.environ[k] = v
    wl = make_workload(name)
    ctx = api.Context(0, ring=wl.ring)
    ctx.load_ccs(wl)
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
    return wl, ctx, scheme


def _oracle(ring):
    if ring == "goldilocks":
        import lfo as O
    else:
        import lfo_bb as O
    return O


@pytest.mark.parametrize("name", ["T8", "T10", "G5", "E22", "E99", "T14", "B6", "B10", "BDP", "B21", "B32", "B14", "B13", "BK8", "B333"])
def test_digit_plane_commits_match_the_oracle(name):
    """both rings: the specialised-wave kernels (13 row tiles: the 25-row chunks of E99; BabyBear kappa 13..16 -> 4 row tiles: B14, B13, BK8, B333), the generic
    guarded ones (everything else), row chunks (E99: 4 x 25 rows; B21 / B32: 2 chunks) and plane groups (E22: 16 + 5; BabyBear: 8 + 7)"""
    wl0 = make_workload(name)
    O = _oracle(wl0.ring)
    tr = lambda: api.PoseidonTranscript(ring=wl0.ring)
    out = {}
    try:
        for mode, env in (("default", None), ("wgs5", {"LF_I8_WGS": "5"}), ("uncoupled", {"LF_I8_COUPLE_W": "0", "LF_I8_WGS": "48"})):
            wl, ctx, scheme = _setup(name, env)
            wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
            cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
            out[mode] = api.LFDecompositionProver.prove(ctx, acc, wit, tr())
            ctx.close()
        inst = O.Instance(wl)
        f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
        acc_o, _ = inst.linearize(O.Transcript(), cccs, f_coeff)
        want = inst.decomposition_prove(O.Transcript(), wl.ajtai_matrix(), acc_o, f_coeff)
        for mode in out:
            assert (out[mode][1] == want[1]).all() and (out[mode][0] == want[0]).all(), mode
    finally:
        for k in ("LF_I8_WGS", "LF_I8_COUPLE_W"):
            os.environ.pop(k, None)


@pytest.mark.parametrize("pattern", ["plus", "minus", "alternating", "random"])
@pytest.mark.parametrize("wgs", ["1", "3", "256"])
def test_extreme_digit_patterns(pattern, wgs):
    """witnesses whose digit planes are saturated: every int8 product has the same sign, so the int32 tile sums and the byte-bias correction
    see their largest magnitudes"""
    import lfo
    try:
        wl, ctx, scheme = _setup("T10", {"LF_I8_WGS": wgs})
        B = wl.B
        n_w = wl.w_ccs.shape[0]
        full = (B // 2 - 1) * sum(B ** l for l in range(wl.L))       # every base-B digit has all its low bits set
        coeff = np.zeros((n_w, 24), dtype=np.int64)
        if pattern == "plus":
            coeff[:] = full
        elif pattern == "minus":
            coeff[:] = -full
        elif pattern == "alternating":
            coeff[:] = full
            coeff[1::2] *= -1
            coeff[:, 1::2] *= -1
        else:
            rng = np.random.default_rng(5)
            coeff = rng.integers(-full, full, size=(n_w, 24))
        P = api.P
        w = lfo.crt(np.array([[int(v) % P for v in row] for row in coeff], dtype=np.uint64))
        wit = api.Witness.from_w_ccs(ctx, w)
        # the decomposition itself does not need a satisfied CCS: take the LCCCS of the reference witness and decompose the saturated one under it
        wit0 = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs0 = np.concatenate([wit0.commit(scheme), wl.x_ccs])
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs0, wit0, api.PoseidonTranscript())
        got = api.LFDecompositionProver.prove(ctx, acc, wit, api.PoseidonTranscript())
        ctx.close()
        inst = lfo.Instance(wl)
        want = inst.decomposition_prove(lfo.Transcript(), wl.ajtai_matrix(), acc, inst.witness_from_w_ccs(w))
        assert (got[1] == want[1]).all() and (got[0] == want[0]).all()
    finally:
        os.environ.pop("LF_I8_WGS", None)
