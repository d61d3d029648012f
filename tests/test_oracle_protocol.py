"""Style-A self-consistency of the oracle, as in the reference's nifs/tests.rs:58-117:
prove one fold step, then the restated NIFSVerifier must accept; tampering must reject."""
import numpy as np
import pytest

import lfo
from latticefold_amd.workload import P, RE, make_workload


def setup_instance(name, seed=0):
    wl = make_workload(name, seed)
    inst = lfo.Instance(wl)
    A = wl.ajtai_matrix()
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    f_ntt = lfo.crt(f_coeff)
    cm = lfo.ajtai_commit(A, wl.kappa, wl.N, f_ntt)
    cccs = np.concatenate([cm, wl.x_ccs])
    # accumulator = linearization of the same instance (benches/utils.rs:619-680)
    acc, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    return wl, inst, A, f_coeff, cccs, acc


@pytest.mark.parametrize("name", ["T8", "G5"])
def test_fold_step_verifies(name):
    wl, inst, A, f_coeff, cccs, acc = setup_instance(name)
    lc, f0, proof = inst.fold_step(lfo.Transcript(), A, acc, f_coeff, cccs, f_coeff)
    rc, lc_v = inst.verify(lfo.Transcript(), acc, cccs, proof)
    assert rc == 0
    assert (lc_v == lc).all()
    # folded witness opens the folded commitment: cm_0 == A * f_0
    cm0 = lc[wl.s + 3: wl.s + 3 + wl.kappa]
    assert (lfo.ajtai_commit(A, wl.kappa, wl.N, f0) == cm0).all()
    # norm of the folded witness stays below B/2 (centred), i.e. it can be folded again
    c = lfo.icrt(f0).astype(object)
    c = np.where(c > P // 2, c - P, c)
    assert max(abs(int(v)) for v in c.reshape(-1)) < wl.B // 2
    # tampered proofs are rejected (decomposition/tests/mod.rs:489-522 style)
    for pos in (0, inst.proof_len // 2, inst.proof_len - 1):
        bad = proof.copy(); bad[pos, 0] = (int(bad[pos, 0]) + 1) % P
        rc, _ = inst.verify(lfo.Transcript(), acc, cccs, bad)
        assert rc != 0


def test_fold_is_deterministic_and_seed_sensitive():
    wl, inst, A, f_coeff, cccs, acc = setup_instance("T8", seed=0)
    a = inst.fold_step(lfo.Transcript(), A, acc, f_coeff, cccs, f_coeff)
    b = inst.fold_step(lfo.Transcript(), A, acc, f_coeff, cccs, f_coeff)
    assert all((x == y).all() for x, y in zip(a, b))
    wl2, inst2, A2, f2, cccs2, acc2 = setup_instance("T8", seed=1)
    c = inst2.fold_step(lfo.Transcript(), A2, acc2, f2, cccs2, f2)
    assert not (a[2] == c[2]).all()
