"""R1CS -> CCS front-end (latticefold_amd/ccs.py = CCS::from_r1cs_padded, arith.rs:122-172) on the reference's own test circuit
x^3 + x + 5 = y (arith/r1cs.rs:128-151): shapes, padding and a full prove -> verify on the CPU oracle; the product's host
verifier must agree.  The GPU versions are in tests/test_gpu_parity.py / test_gpu_bb.py."""
import numpy as np
import pytest

from latticefold_amd import api, ccs


def build(ring):
    A, B, C = ccs.vitalik_r1cs()
    z = ccs.vitalik_z_ntt(ring)
    assert ccs.check_r1cs_scalar_slots(A, B, C, z, ring)
    x_ccs, w_ccs = z[0:1], z[2:]
    if ring == "goldilocks":
        return ccs.workload_from_r1cs(A, B, C, 1, x_ccs, w_ccs, ring=ring, L=4, Bbase=1 << 16, K=16, kappa=4, name="vitalik")
    return ccs.workload_from_r1cs(A, B, C, 1, x_ccs, w_ccs, ring=ring, L=2, Bbase=1 << 16, K=16, kappa=3, name="vitalik_bb")


@pytest.mark.parametrize("ring", ["goldilocks", "babybear"])
def test_from_r1cs_padded_shapes(ring):
    wl = build(ring)
    assert wl.n == 6 and wl.wit_len == 4 and wl.t == 3 and wl.q == 2 and wl.d == 2
    assert wl.m == max(wl.wit_len * wl.L, 4) and wl.m & (wl.m - 1) == 0      # max((n-l-1)*L, m).next_power_of_two()
    for rp in wl.rowptr:
        assert len(rp) == wl.m + 1 and (np.diff(rp.astype(np.int64))[4:] == 0).all()   # padded rows are empty
    assert (wl.z()[1] == ccs.diag(1, ring)).all()                           # z = x || 1 || w


@pytest.mark.parametrize("ring", ["goldilocks", "babybear"])
def test_fold_real_circuit_on_the_oracle(ring):
    if ring == "goldilocks":
        import lfo as o
    else:
        import lfo_bb as o
    wl = build(ring)
    inst = o.Instance(wl)
    A = wl.ajtai_matrix()
    f = inst.witness_from_w_ccs(wl.w_ccs)
    cm = o.ajtai_commit(A, wl.kappa, wl.N, o.crt(f))
    cccs = np.concatenate([cm, wl.x_ccs])
    acc, _ = inst.linearize(o.Transcript(), cccs, f)
    lc, f0, pr = inst.fold_step(o.Transcript(), A, acc, f, cccs, f)
    rc, lc2 = inst.verify(o.Transcript(), acc, cccs, pr)
    assert rc == 0 and (lc == lc2).all()            # satisfied circuit -> accepted
    ok, lc3, _ = api.NIFSVerifier.verify(wl, acc, cccs, pr, api.PoseidonTranscript(ring=ring))
    assert ok and (lc3 == lc).all()
    # an assignment that violates the circuit (y off by one) is rejected at the linearization sumcheck
    bad = wl.w_ccs.copy(); bad[0, 0] = (int(bad[0, 0]) + 1) % wl.P
    fb = inst.witness_from_w_ccs(bad)
    cmb = o.ajtai_commit(A, wl.kappa, wl.N, o.crt(fb))
    cccsb = np.concatenate([cmb, wl.x_ccs])
    accb, _ = inst.linearize(o.Transcript(), cccsb, fb)
    _, _, prb = inst.fold_step(o.Transcript(), A, accb, fb, cccsb, fb)
    ok, _, stage = api.NIFSVerifier.verify(wl, accb, cccsb, prb, api.PoseidonTranscript(ring=ring))
    assert not ok and stage == 1
