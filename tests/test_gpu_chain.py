"""Chained folding (IVC style) through the C ABI: every step ingests a NEW witness of the same constraint system (Witness::from_w_ccs, arith.rs:230-248),
commits it (Witness::commit, arith.rs:357-362 -- the int8 general commit), and folds it into the carried accumulator (NIFSProver::prove, nifs.rs:48-103;
chaining as in nifs/tests.rs:58-117).  T14 / C2 / B10: every step against the committed oracle-only digests (tests/tools/make_chain_digests.py); C4: size-independent
properties per step (the oracle's restated verifier accepts, the folded witness opens the folded commitment, its norm stays below B/2)."""
import hashlib
import json
import os

import numpy as np
import pytest

from latticefold_amd import api
from latticefold_amd.workload import chain_w_ccs, make_workload

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_digests.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


@pytest.mark.parametrize("name,overlap", [("T14", False), ("T14", True), ("C2", False), ("C2", True), ("B10", False), ("B10", True)])
def test_chain_matches_the_oracle_step_by_step(name, overlap):
    """overlap: the witness of step j + 1 is ingested while step j folds (lf_witness_from_w_ccs_begin / lf_witness_job_finish: lane 2 of a Goldilocks context, the
    blocking call on a worker thread for BabyBear) -- the same digests"""
    gold = GOLD[name]
    wl = make_workload(name)
    ctx = api.Context(0, ring=wl.ring)
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        tr = lambda: api.PoseidonTranscript(ring=wl.ring)
        w_acc = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        acc, _ = api.LFLinearizationProver.prove(ctx, np.concatenate([w_acc.commit(scheme), wl.x_ccs]), w_acc, tr())
        assert sha(acc) == gold["acc0"]
        nsteps = len(gold["steps"])
        pending = api.Witness.from_w_ccs_begin(ctx, chain_w_ccs(wl, 1)) if overlap else None
        for j, g in enumerate(gold["steps"], start=1):
            w = chain_w_ccs(wl, j)
            assert sha(w) == g["w_ccs"]
            w_j = pending.result() if overlap else api.Witness.from_w_ccs(ctx, w)
            cm = w_j.commit(scheme)
            assert sha(cm) == g["cm"], (j, "cm")
            if overlap and j < nsteps:
                pending = api.Witness.from_w_ccs_begin(ctx, chain_w_ccs(wl, j + 1))
            lc, w_next, proof = api.NIFSProver.prove(ctx, acc, w_acc, np.concatenate([cm, wl.x_ccs]), w_j, tr())
            assert sha(proof) == g["proof"], (j, "proof")
            assert sha(lc) == g["lcccs"], (j, "lcccs")
            assert sha(w_next.f) == g["f_ntt"], (j, "f_ntt")
            ok, mx = ctx.linf_check(w_next.f, wl.B // 2)
            assert ok and mx == g["norm"], (j, mx, g["norm"])
            w_j.free()
            w_acc.free()
            acc, w_acc = lc, w_next
    finally:
        ctx.close()


def test_overlapped_ingestion_equals_the_blocking_one_and_can_be_abandoned():
    """C2: two jobs in flight are serialised by the library, the witnesses equal Witness::from_w_ccs word for word; a job that is dropped frees what it made; a job
    on a context without a constraint system is refused"""
    wl = make_workload("C2")
    ctx = api.Context(0)
    try:
        with pytest.raises(api.LfError):
            api.Witness.from_w_ccs_begin(ctx, wl.w_ccs)
        ctx.load_ccs(wl)
        ws = [chain_w_ccs(wl, j) for j in (1, 2, 3)]
        jobs = [api.Witness.from_w_ccs_begin(ctx, w) for w in ws]
        jobs[2].abandon()
        for w, job in zip(ws[:2], jobs[:2]):
            a, b = job.result(), api.Witness.from_w_ccs(ctx, w)
            assert (a.f_coeff == b.f_coeff).all() and (a.w_ccs == w).all()
            a.free(); b.free()
        with pytest.raises(RuntimeError):
            jobs[0].result()
    finally:
        ctx.close()


def test_chain_properties_at_c4():
    import lfo
    wl = make_workload("C4")
    ctx = api.Context(0)
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        inst = lfo.Instance(wl)
        w_acc = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        acc, _ = api.LFLinearizationProver.prove(ctx, np.concatenate([w_acc.commit(scheme), wl.x_ccs]), w_acc, api.PoseidonTranscript())
        for j in (1, 2, 3):
            w_j = api.Witness.from_w_ccs(ctx, chain_w_ccs(wl, j))
            cccs = np.concatenate([w_j.commit(scheme), wl.x_ccs])
            lc, w_next, proof = api.NIFSProver.prove(ctx, acc, w_acc, cccs, w_j, api.PoseidonTranscript())
            rc, lc_v = inst.verify(lfo.Transcript(), acc, cccs, proof)
            assert rc == 0 and (lc_v == lc).all(), j
            assert (w_next.commit(scheme) == lc[wl.s + 3: wl.s + 3 + wl.kappa]).all(), j      # the folded witness opens the folded commitment
            ok, mx = ctx.linf_check(w_next.f, wl.B // 2)
            assert ok, (j, mx)
            w_j.free()
            w_acc.free()
            acc, w_acc = lc, w_next
    finally:
        ctx.close()
