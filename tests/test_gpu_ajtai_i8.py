"""The int8 matrix-core commit kernel (lf_ajtai_i8.hip) against the 64-bit VALU kernel and the oracle: the K-1 digit-plane commitments of
LFDecompositionProver::prove (decomposition.rs:178-201), word for word, over shapes that exercise ragged column tiles (n not a multiple
of 8 * workgroups), odd kappa (padded row tile), row chunks (kappa > 26), plane groups (K - 1 > 16), extreme digits (all +1 / all -1 /
alternating: the int32 accumulators and the -128 byte bias), and a few workgroup counts."""
import os

import numpy as np
import pytest

from latticefold_amd import api
from latticefold_amd.workload import make_workload

pytestmark = pytest.mark.gpu


def _y_s(ctx, wl, wit, acc):
    lcccs_s, proof = api.LFDecompositionProver.prove(ctx, acc, wit, api.PoseidonTranscript())
    return lcccs_s, proof


def _setup(name, valu, env=None):
    for k in ("LF_AJTAI_VALU", "LF_I8_WGS"):
        os.environ.pop(k, None)
    if valu:
        os.environ["LF_AJTAI_VALU"] = "1"
    for k, v in (env or {}).items():
        os.environ[k] = v
    wl = make_workload(name)
    ctx = api.Context(0)
    ctx.load_ccs(wl)
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
    return wl, ctx, scheme


@pytest.mark.parametrize("name", ["T8", "T10", "G5", "E22", "E99"])
def test_digit_plane_commits_match_valu_kernel_and_oracle(name):
    import lfo
    out = {}
    try:
        for valu in (False, True):
            wl, ctx, scheme = _setup(name, valu)
            wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
            cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
            out[valu] = _y_s(ctx, wl, wit, acc)
            ctx.close()
        assert (out[False][0] == out[True][0]).all() and (out[False][1] == out[True][1]).all()
        inst = lfo.Instance(wl)
        f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
        acc_o, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
        want = inst.decomposition_prove(lfo.Transcript(), wl.ajtai_matrix(), acc_o, f_coeff)
        assert (out[False][1] == want[1]).all() and (out[False][0] == want[0]).all()
    finally:
        os.environ.pop("LF_AJTAI_VALU", None)


@pytest.mark.parametrize("pattern", ["plus", "minus", "alternating", "random"])
@pytest.mark.parametrize("wgs", ["1", "3", "256"])
def test_extreme_digit_patterns(pattern, wgs):
    """witnesses whose digit planes are saturated: every int8 product has the same sign, so the int32 tile sums and the byte-bias correction
    see their largest magnitudes"""
    out = {}
    try:
        for valu in (False, True):
            wl, ctx, scheme = _setup("T10", valu, None if valu else {"LF_I8_WGS": wgs})
            B = wl.B
            n_w = wl.w_ccs.shape[0]
            import lfo
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
            cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
            # the decomposition itself does not need a satisfied CCS: take the LCCCS of the reference witness and swap the commitment in
            wit0 = api.Witness.from_w_ccs(ctx, wl.w_ccs)
            cccs0 = np.concatenate([wit0.commit(scheme), wl.x_ccs])
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs0, wit0, api.PoseidonTranscript())
            out[valu] = api.LFDecompositionProver.prove(ctx, acc, wit, api.PoseidonTranscript())
            ctx.close()
        assert (out[False][1] == out[True][1]).all()
    finally:
        for k in ("LF_AJTAI_VALU", "LF_I8_WGS"):
            os.environ.pop(k, None)
