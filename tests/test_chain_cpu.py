"""The witnesses of the chained-folding workload (workload.chain_w_ccs; bench.py --chain, tests/test_gpu_chain.py) on the host: they satisfy the workload's fixed
constraint system, and the committed oracle-only chain digests (tests/golden/chain_digests.json, tests/tools/make_chain_digests.py) are made from them; the oracle
replays the first step of the smallest chain."""
import hashlib
import json
import os

import numpy as np

from latticefold_amd.workload import chain_w_ccs, make_workload

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chain_digests.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def test_chain_witnesses_satisfy_the_fixed_constraint_system():
    """(host) z_j * z_j == z_base * z_j slot by slot: the R1CS A = B = I, C = diag(z_base) accepts every chain witness"""
    from latticefold_amd.workload import ring_mul_ntt
    for name in ("T8", "B6"):
        wl = make_workload(name)
        for j in (1, 2, 5):
            w = chain_w_ccs(wl, j)
            assert (ring_mul_ntt(w, w, wl.ring) == ring_mul_ntt(wl.w_ccs, w, wl.ring)).all()
            assert 0 < (w != wl.w_ccs).sum() and (w != 0).any()



def test_chain_fixture_inputs_and_norms():
    for name, g in GOLD.items():
        wl = make_workload(name)
        for j, st in enumerate(g["steps"], start=1):
            assert sha(chain_w_ccs(wl, j)) == st["w_ccs"]
            assert st["norm"] < wl.B // 2


def test_oracle_replays_the_first_chain_step():
    import lfo
    wl = make_workload("T10")
    inst = lfo.Instance(wl)
    A = inst.ajtai_matrix()
    f_acc = inst.witness_from_w_ccs(wl.w_ccs)
    acc, _ = inst.linearize(lfo.Transcript(), np.concatenate([lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_acc)), wl.x_ccs]), f_acc)
    f_1 = inst.witness_from_w_ccs(chain_w_ccs(wl, 1))
    cccs = np.concatenate([lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_1)), wl.x_ccs])
    lc, f0, proof = inst.fold_step(lfo.Transcript(), A, acc, f_acc, cccs, f_1)
    rc, lc_v = inst.verify(lfo.Transcript(), acc, cccs, proof)
    assert rc == 0 and (lc_v == lc).all()
