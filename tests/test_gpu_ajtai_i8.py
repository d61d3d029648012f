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
    for k in ("LF_AJTAI_VALU", "LF_I8_WGS", "LF_I8_GUARDED", "LF_I8_COLS", "LF_I8_BITS"):
        os.environ.pop(k, None)
    if valu:
        os.environ["LF_AJTAI_VALU"] = "1"
    for k, v in (env or {}).items():
        os.environ[k] = v
    wl = make_workload(name)
    ctx = api.Context(0, ring=wl.ring)
    ctx.load_ccs(wl)
    scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
    return wl, ctx, scheme


@pytest.mark.parametrize("name", ["T8", "T10", "G5", "E22", "E99", "T14", "B6", "B10", "BDP", "B21", "B32", "B14", "B13", "BK8", "B333"])
def test_digit_plane_commits_match_valu_kernel_and_oracle(name):
    """both rings: the exact-count instantiations (13 row tiles: the 25-row chunks of E99; BabyBear kappa 13..16 -> 4 row tiles: B14),
    the generic guarded ones (everything else), row chunks (E99: 4 x 25 rows; B21 / B32: 2 chunks) and plane groups (E22: 16 + 5; BabyBear: 8 + 7)"""
    wl0 = make_workload(name)
    if wl0.ring == "goldilocks":
        import lfo as O
    else:
        import lfo_bb as O
    tr = lambda: api.PoseidonTranscript(ring=wl0.ring)
    out = {}
    try:
        # nocols / nobits: the specialised 24-ring kernel (13-row-tile shapes) with the 2 x 2 split of its multiplier waves / with digits cut from the int32 planes
        envs = {"guarded": {"LF_I8_GUARDED": "1"}, "nocols": {"LF_I8_COLS": "0"}, "nobits": {"LF_I8_BITS": "0", "LF_I8_COLS": "0"}}
        for mode in ("i8", "valu", "guarded", "nocols", "nobits"):
            wl, ctx, scheme = _setup(name, mode == "valu", envs.get(mode))
            wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
            cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
            out[mode] = api.LFDecompositionProver.prove(ctx, acc, wit, tr())
            ctx.close()
        for mode in ("valu", "guarded", "nocols", "nobits"):
            assert (out["i8"][0] == out[mode][0]).all() and (out["i8"][1] == out[mode][1]).all(), mode
        inst = O.Instance(wl)
        f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
        acc_o, _ = inst.linearize(O.Transcript(), cccs, f_coeff)
        want = inst.decomposition_prove(O.Transcript(), wl.ajtai_matrix(), acc_o, f_coeff)
        assert (out["i8"][1] == want[1]).all() and (out["i8"][0] == want[0]).all()
    finally:
        for k in ("LF_AJTAI_VALU", "LF_I8_GUARDED", "LF_I8_COLS", "LF_I8_BITS"):
            os.environ.pop(k, None)


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


@pytest.mark.parametrize("name", ["T8", "T10", "G5", "E22", "E99", "T14"])
def test_paired_commit_of_both_decompositions(name):
    """LF_I8_PAIR: lf_fold_step commits the digit planes of BOTH witnesses in one pass over A (paired workgroups, lf_ajtai_i8.hip `sides`):
    the chained step -- left witness = the folded witness of the first step, right witness = w_i, so the two sides differ -- must equal the
    default run with one launch per decomposition for several workgroup counts (odd chunk counts, empty trailing chunks), and the oracle."""
    import lfo
    out = {}
    try:
        for mode, env in (("pair", {"LF_I8_PAIR": "1"}), ("nopair", None), ("pair16", {"LF_I8_PAIR": "1", "LF_I8_WGS": "16"}),
                          ("pair40", {"LF_I8_PAIR": "1", "LF_I8_WGS": "40"}), ("pair250", {"LF_I8_PAIR": "1", "LF_I8_WGS": "250"})):
            os.environ.pop("LF_I8_PAIR", None)
            wl, ctx, scheme = _setup(name, False, env)
            wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
            cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript())
            lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, api.PoseidonTranscript())
            lc2, w1, proof2 = api.NIFSProver.prove(ctx, lc, w0, cccs, wit, api.PoseidonTranscript())
            out[mode] = (proof, lc, proof2, lc2, w1.f)
            ctx.close()
        for mode in out:
            for a, b in zip(out["nopair"], out[mode]):
                assert (a == b).all(), mode
        if name in ("T8", "T10", "G5"):
            inst = lfo.Instance(wl)
            f = inst.witness_from_w_ccs(wl.w_ccs)
            A = wl.ajtai_matrix()
            acc_o, _ = inst.linearize(lfo.Transcript(), cccs, f)
            lc_o, f0_o, proof_o = inst.fold_step(lfo.Transcript(), A, acc_o, f, cccs, f)
            lc2_o, f1_o, proof2_o = inst.fold_step(lfo.Transcript(), A, lc_o, lfo.icrt(f0_o), cccs, f)
            assert (out["pair"][2] == proof2_o).all() and (out["pair"][3] == lc2_o).all() and (out["pair"][4] == f1_o).all()
    finally:
        for k in ("LF_I8_PAIR", "LF_I8_WGS"):
            os.environ.pop(k, None)


@pytest.mark.parametrize("name,how", [("T10", "matrix"), ("E99", "matrix"), ("T14", "seed"), ("G5", "seed")])
def test_digits_only_context_matches_the_full_one(name, how):
    """lf_ajtai_set_digits_only: the context keeps only the byte planes of A (rows pass through one u64 row buffer: fused inverse map +
    packing).  Digit-plane commitments (the fold step), general commitments (NTT form rebuilt from the bytes for the call) and the
    witness commitment must be word for word those of a context that holds both forms; switching the mode on afterwards drops the copy."""
    for k in ("LF_AJTAI_VALU", "LF_I8_WGS", "LF_I8_GUARDED", "LF_I8_COLS", "LF_I8_BITS"):
        os.environ.pop(k, None)
    wl = make_workload(name)
    from latticefold_amd.workload import splitmix_fq
    f = splitmix_fq(99, 0, 2 * wl.N * 24).reshape(2, wl.N, 24)
    out = {}
    for mode in ("full", "digits", "late"):
        ctx = api.Context(0, ring=wl.ring)
        ctx.load_ccs(wl)
        kw = dict(matrix=wl.ajtai_matrix()) if how == "matrix" else dict(kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        scheme = api.AjtaiCommitmentScheme(ctx, digits_only=(mode == "digits"), **kw)
        if mode == "late":
            free0 = ctx.device_memory()[0]
            scheme.set_digits_only(True)
            copy_bytes = wl.kappa * wl.N * 24 * 8
            if copy_bytes >= 32 << 20:   # (small buffers come out of the runtime's own sub-allocator: nothing to see in mem_get_info)
                assert ctx.device_memory()[0] - free0 >= copy_bytes * 0.9      # the NTT-form copy went back to the driver
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cm = wit.commit(scheme)
        cccs = np.concatenate([cm, wl.x_ccs])
        tr = api.PoseidonTranscript(ring=wl.ring)
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr)
        dec = api.LFDecompositionProver.prove(ctx, acc, wit, tr)
        out[mode] = (cm, scheme.commit_ntt(f[0]), dec[0], dec[1], scheme.commit_ntt(f[1]))
        ctx.close()
    for mode in ("digits", "late"):
        for a, b in zip(out["full"], out[mode]):
            assert (np.asarray(a) == np.asarray(b)).all(), mode
