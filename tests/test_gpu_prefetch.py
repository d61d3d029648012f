"""lf_prefetch_instance (include/lfhip.h): the hint that lets a fold step prepare the challenge-independent half of the NEXT step's right decomposition
(nifs/decomposition.rs:159-201 -- bit planes, z_k = x_s[k] || w_k, the K - 1 digit-plane commitments of w_i) while its own launches are latency-bound.
Proofs must be bit-identical with and without the hint, against the CPU oracle, wherever the step enqueues the work (LF_PF_AT); a result is used exactly
once and only by the witness (handle AND serial number) and public input it was made for."""
import numpy as np
import pytest

import lfo
from latticefold_amd import api
from latticefold_amd.workload import make_workload

pytestmark = pytest.mark.gpu


def _instances(ctx, name, scheme, count):
    """`count` fresh instances over the CCS of seed 0 (parity does not need the relation to hold: the prover never checks it)"""
    wits, cccs, fs = [], [], []
    for k in range(count):
        wk = make_workload(name, seed=k)
        w = api.Witness.from_w_ccs(ctx, wk.w_ccs)
        wits.append(w)
        cccs.append(np.concatenate([w.commit(scheme), wk.x_ccs]))
        fs.append(wk)
    return wits, cccs, fs


def _chain(ctx, wl, scheme, wits, cccs, steps, hint):
    """acc_0 = linearized instance 0; step i folds (acc, instance i).  hint(i) is called before step i (i = 1..steps)."""
    tr = api.PoseidonTranscript()
    acc, _ = api.LFLinearizationProver.prove(ctx, cccs[0], wits[0], tr)
    w_acc, out, made = wits[0], [], []
    for i in range(1, steps + 1):
        hint(i)
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, w_acc, cccs[i], wits[i], tr)
        out.append((lc, proof, w0.f))
        acc, w_acc = lc, w0
        made.append(w0)
    for w in made:
        w.free()
    return out


@pytest.mark.parametrize("name,pf_at", [("T10", 0), ("T10", (0, 2)), ("T10", (2, 50)), ("T12", 0), ("T12", 1), ("T12", 2), ("T12", 3), ("T12", 13), ("T12", 16), ("T12", 40), ("T12", 50),
                                        ("T12", (0, 17)), ("T12", (2, 16)), ("T12", (2, 40)), ("T12", (12, 50)), ("T14", 0), ("T14", 17), ("T14", (2, 17))])
def test_prefetched_chain_is_bit_identical_and_matches_oracle(name, pf_at, monkeypatch):
    # LF_PF_AT: where the step enqueues the bit planes and z_k of the next right side; LF_PF_AT2 (default: the same point): where it enqueues the commits
    if isinstance(pf_at, tuple):
        monkeypatch.setenv("LF_PF_AT2", str(pf_at[1]))
        pf_at = pf_at[0]
    else:
        monkeypatch.delenv("LF_PF_AT2", raising=False)
    monkeypatch.setenv("LF_PF_AT", str(pf_at))
    wl = make_workload(name)
    ctx = api.Context(0)
    try:
        ctx.load_ccs(wl)
        A = wl.ajtai_matrix()
        scheme = api.AjtaiCommitmentScheme(ctx, matrix=A)
        steps = 3
        wits, cccs, wks = _instances(ctx, name, scheme, steps + 1)
        plain = _chain(ctx, wl, scheme, wits, cccs, steps, lambda i: None)
        assert ctx.prefetch_stats() == (0, 0, 0)
        # before step i: announce the instance of step i + 1 (nothing to announce before the last step)
        hinted = _chain(ctx, wl, scheme, wits, cccs, steps, lambda i: ctx.prefetch_instance(cccs[i + 1], wits[i + 1]) if i < steps else None)
        issued, used, dropped = ctx.prefetch_stats()
        assert (issued, used, dropped) == (steps - 1, steps - 1, 0)
        for (lc_a, pr_a, f_a), (lc_b, pr_b, f_b) in zip(plain, hinted):
            assert (pr_a == pr_b).all() and (lc_a == lc_b).all() and (f_a == f_b).all()
        # ... and the chain equals the oracle's (the first two steps: the second one is the first that consumes a prefetch)
        inst = lfo.Instance(wl)
        f = [inst.witness_from_w_ccs(wk.w_ccs) for wk in wks]
        to = lfo.Transcript()
        acc_o, _ = inst.linearize(to, cccs[0], f[0])
        f_acc = f[0]
        for i in (1, 2):
            lc_o, f0_o, proof_o = inst.fold_step(to, A, acc_o, f_acc, cccs[i], f[i])
            lc_g, pr_g, f_g = hinted[i - 1]
            assert (pr_g == proof_o).all() and (lc_g == lc_o).all() and (f_g == f0_o).all(), f"step {i}"
            acc_o, f_acc = lc_o, lfo.icrt(f0_o)
    finally:
        ctx.close()


def test_stale_prefetch_is_rejected():
    """A result made for another witness, for a freed-and-reallocated handle, or for another public input is dropped: the step computes its right side itself and
    the proof is the one of a context that never saw a hint"""
    name = "T12"
    wl = make_workload(name)
    ctx = api.Context(0)
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
        wits, cccs, _ = _instances(ctx, name, scheme, 4)
        ref = _chain(ctx, wl, scheme, wits, cccs, 3, lambda i: None)

        # (a) announced instance 3 for step 2, but step 2 folds instance 2
        got = _chain(ctx, wl, scheme, wits, cccs, 3, lambda i: ctx.prefetch_instance(cccs[3], wits[3]) if i == 1 else None)
        assert ctx.prefetch_stats() == (1, 0, 1)
        for a, b in zip(ref, got):
            assert all((x == y).all() for x, y in zip(a, b))

        # (b) the right witness, another public input x_ccs
        other = cccs[2].copy()
        other[wl.kappa:, 0] ^= np.uint64(1)
        got = _chain(ctx, wl, scheme, wits, cccs, 3, lambda i: ctx.prefetch_instance(other, wits[2]) if i == 1 else None)
        assert ctx.prefetch_stats() == (2, 0, 2)
        for a, b in zip(ref, got):
            assert all((x == y).all() for x, y in zip(a, b))

        # (c) the announced handle is freed and a new witness takes its place (possibly at the same address): another serial number
        w2 = make_workload(name, seed=2).w_ccs
        doomed = api.Witness.from_w_ccs(ctx, make_workload(name, seed=9).w_ccs)
        alt = list(wits)

        def hint(i):
            if i == 1:
                ctx.prefetch_instance(cccs[2], doomed)
            if i == 2:                      # step 1 has enqueued the work for `doomed`; now it dies and the real instance 2 is created afresh
                doomed.free()
                alt[2] = api.Witness.from_w_ccs(ctx, w2)
        tr = api.PoseidonTranscript()
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs[0], wits[0], tr)
        w_acc, got = wits[0], []
        for i in range(1, 4):
            hint(i)
            lc, w0, proof = api.NIFSProver.prove(ctx, acc, w_acc, cccs[i], alt[i], tr)
            got.append((lc, proof, w0.f))
            acc, w_acc = lc, w0
        assert ctx.prefetch_stats() == (3, 0, 3)
        for a, b in zip(ref, got):
            assert all((x == y).all() for x, y in zip(a, b))

        # (d) a request that no step takes up does not outlive the next step; a second request replaces the first
        ctx.prefetch_instance(cccs[1], wits[1])
        ctx.prefetch_instance(cccs[2], wits[2])
        assert ctx.prefetch_stats() == (3, 0, 4)
        with pytest.raises(api.LfError):
            api._chk(api._lib().lf_prefetch_instance(ctx.h, None, wits[1].h), "null cm")
    finally:
        ctx.close()


def test_hint_is_ignored_where_no_prefetch_path_exists(monkeypatch):
    """BabyBear contexts and VALU commits accept the hint and do nothing with it"""
    wl = make_workload("B6")
    ctx = api.Context(0, ring="babybear")
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
        w = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        ctx.prefetch_instance(np.concatenate([w.commit(scheme), wl.x_ccs]), w)
        assert ctx.prefetch_stats() == (0, 0, 0)
    finally:
        ctx.close()
    monkeypatch.setenv("LF_AJTAI_VALU", "1")
    wl = make_workload("T10")
    ctx = api.Context(0)
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
        wits, cccs, _ = _instances(ctx, "T10", scheme, 3)
        ref = _chain(ctx, wl, scheme, wits, cccs, 2, lambda i: None)
        got = _chain(ctx, wl, scheme, wits, cccs, 2, lambda i: ctx.prefetch_instance(cccs[2], wits[2]) if i == 1 else None)
        assert ctx.prefetch_stats() == (0, 0, 1)
        for a, b in zip(ref, got):
            assert all((x == y).all() for x, y in zip(a, b))
    finally:
        ctx.close()
