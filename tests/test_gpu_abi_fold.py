"""SURVEY 8(b) exports that a Rust shim would bind at the folding prover's call sites, against the oracle on random data, both rings:
  lf_sumcheck_fold_{begin,round,end}  MLSumcheck::prove_as_subprotocol (utils/sumcheck.rs:53-80), comb nifs/folding/utils.rs:273-325
  lf_horner_combine                   calculate_challenged_mz_mle (nifs/folding.rs:208-226), utils.rs:524-546
  lf_lincomb                          compute_f_0 (nifs/folding.rs:258-268)
"""
import numpy as np
import pytest

from latticefold_amd import api
from latticefold_amd.workload import diag, make_workload, splitmix_fq

pytestmark = pytest.mark.gpu


def _mods(ring):
    if ring == "goldilocks":
        import lfo as O
    else:
        import lfo_bb as O
    return O


def _embed(ch, wl):
    """F_{p^tau} challenge (tau words) -> diagonal ring element"""
    return np.tile(np.asarray(ch, dtype=np.uint64), 8)


def _rand_ring(seed, n, wl):
    return splitmix_fq(seed, 0, n * wl.RE, wl.ring).reshape(n, wl.RE)


@pytest.mark.parametrize("name", ["T8", "B6"])
def test_sumcheck_fold_abi_matches_oracle(name):
    wl = make_workload(name)
    O = _mods(wl.ring)
    ctx = api.Context(0, ring=wl.ring)
    try:
        ctx.load_ccs(wl)
        m, tau, K2 = wl.m, wl.tau, 2 * wl.K
        P = 5 + K2 * tau
        tables = _rand_ring(77, P * m, wl).reshape(P, m, wl.RE)
        # eq tables are slot-constant in the protocol: real eq tables of random points
        for idx, sd in ((0, 1), (2, 2), (4, 3)):
            pt = splitmix_fq(100 + sd, 0, wl.s * tau, wl.ring).reshape(wl.s, tau)
            tables[idx] = O.build_eq(np.stack([_embed(c, wl) for c in pt]))
        # f-hat tables: small digits like the protocol's (-1, 0, 1) mixed with a few arbitrary values (the comb is generic)
        dig = (splitmix_fq(5, 0, K2 * tau * m * wl.RE, wl.ring) % np.uint64(3)).reshape(K2 * tau, m, wl.RE)
        fh = np.where(dig == 2, np.uint64(wl.P - 1), dig).astype(np.uint64)
        fh[::7] = tables[5::7][: fh[::7].shape[0]]
        tables[5:] = fh
        mu = splitmix_fq(9, 0, K2 * tau, wl.ring).reshape(K2, tau)
        mu[-1] = 0
        mu[-1, 0] = 1
        inst = O.Instance(wl)
        msgs_o, pt_o = inst.sumcheck_fold(O.Transcript(), tables, np.stack([_embed(c, wl) for c in mu]))
        sc = api.MLSumcheckFold(ctx, tables, mu)
        with pytest.raises(api.LfError):
            sc.prove_round(pt_o[0][:tau])            # first round takes no verifier message
        npts = 2 * wl.b + 1
        for rnd in range(wl.s):
            ev = sc.prove_round(None if rnd == 0 else pt_o[rnd - 1][:tau])
            assert (ev == msgs_o[rnd * npts:(rnd + 1) * npts]).all(), f"round {rnd + 1}"
        with pytest.raises(api.LfError):
            sc.prove_round(pt_o[-1][:tau])           # "Prover is not active"
        sc.end()
        # non-slot-constant eq table is refused, not mis-evaluated
        bad = tables.copy()
        bad[0, 3, 1] ^= np.uint64(1)
        with pytest.raises(api.LfError):
            api.MLSumcheckFold(ctx, bad, mu)
    finally:
        ctx.close()


@pytest.mark.parametrize("ring,tau", [("goldilocks", 3), ("babybear", 9)])
def test_horner_combine_and_lincomb_match_oracle(ring, tau):
    wl = make_workload("T8" if ring == "goldilocks" else "B6")
    O = _mods(ring)
    ctx = api.Context(0, ring=ring)
    try:
        for groups, per_group, ln in ((1, 1, 1), (3, 2, 37), (32, 3, 300), (5, 9, 64)):
            t = _rand_ring(groups * 1000 + ln, groups * per_group * ln, wl).reshape(groups, per_group, ln, wl.RE)
            ch = splitmix_fq(groups + 17, 0, groups * tau, ring).reshape(groups, tau)
            want = O.horner_combine(t, np.stack([_embed(c, wl) for c in ch]))
            assert (api.horner_combine(ctx, t, ch) == want).all(), (groups, per_group, ln)
        for n, ln in ((1, 1), (2, 5), (32, 257), (7, 1000)):
            t = _rand_ring(n * 31 + ln, n * ln, wl).reshape(n, ln, wl.RE)
            cf = _rand_ring(n + 5, n, wl)           # genuine ring elements: 8 distinct slots (rho_i = CRT of a short challenge)
            assert (api.lincomb(ctx, cf, t) == O.lincomb(cf, t)).all(), (n, ln)
        with pytest.raises(api.LfError):
            api.lincomb(ctx, np.zeros((0, wl.RE), dtype=np.uint64), np.zeros((0, 4, wl.RE), dtype=np.uint64))
    finally:
        ctx.close()
