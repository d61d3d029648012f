"""Generates tests/golden/scale_digests.json: SHA-256 digests of one complete fold step (`NIFSProver::prove`) at the BASELINE sizes,
computed by the CPU ORACLE ONLY (oracle/liblfo*.so -- no GPU, no product code beyond the numpy workload generator).

    python tests/tools/make_scale_digests.py [names...]        default: C2 T18 C4 B14 C3;   name/ccs (C2/multi4, C2/multi16, C2/deg3): general constraint systems

Inputs are the deterministic synthetic workloads of latticefold_amd/workload.py (the same ones bench.py runs): acc = linearization of
the instance under a fresh transcript, then fold_step(acc, w, cm_i, w) under a fresh transcript, exactly the call sequence of the
reference's e2e bench (benches/utils.rs:619-680).  The GPU tests (tests/test_gpu_parity_scale.py) recompute the same objects through
the C ABI and compare digests section by section.  C4 needs ~70 GiB of host RAM and a few minutes on 16 cores; C3 ~10 minutes.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden", "scale_digests.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def sections(wl, acc, lc, f0, proof):
    """digest per protocol section so that a mismatch points at the phase that produced it"""
    tau = wl.tau
    lin = wl.s * (wl.d + 2) + tau + wl.t
    dec = wl.K * (wl.t + tau + wl.l + 1 + wl.kappa)
    fold_msgs = wl.s * (2 * wl.b + 1)
    p = np.asarray(proof).reshape(-1, wl.RE)
    o = 2 * dec + lin
    return {
        "acc": sha(acc), "lcccs_out": sha(lc), "f0_ntt": sha(f0),
        "proof_lin": sha(p[:lin]), "proof_dec_left": sha(p[lin:lin + dec]), "proof_dec_right": sha(p[lin + dec:o]),
        "proof_fold_msgs": sha(p[o:o + fold_msgs]), "proof_theta": sha(p[o + fold_msgs:o + fold_msgs + 2 * wl.K * tau]),
        "proof_eta": sha(p[o + fold_msgs + 2 * wl.K * tau:]), "proof": sha(p),
        "first_words": {"acc": [int(x) for x in np.asarray(acc).reshape(-1)[:3]], "lcccs_out_v0": [int(x) for x in np.asarray(lc).reshape(-1, wl.RE)[wl.s][:3]],
                        "proof_last": [int(x) for x in p[-1][:3]]},
    }


def run(name, ccs="r1cs"):
    """ccs: the constraint-system shape of workload.make_workload ("r1cs": the bench's 1-nnz R1CS; "multi4" / "multi16": 4 / 16 entries per row at random
    columns; "deg3": the reference's degree-three CCS) -- stored under the key name/ccs"""
    from latticefold_amd.workload import make_workload
    wl = make_workload(name, ccs=ccs)
    if wl.ring == "goldilocks":
        import lfo as O
    else:
        import lfo_bb as O
    t0 = time.time()
    inst = O.Instance(wl)
    A = inst.ajtai_matrix()
    f = inst.witness_from_w_ccs(wl.w_ccs)
    cm = O.ajtai_commit(A, wl.kappa, wl.N, O.crt(f))
    cccs = np.concatenate([cm, wl.x_ccs])
    acc, _ = inst.linearize(O.Transcript(), cccs, f)
    t1 = time.time()
    lc, f0, proof = inst.fold_step(O.Transcript(), A, acc, f, cccs, f)
    t2 = time.time()
    d = sections(wl, acc, lc, f0, proof)
    d["oracle_seconds"] = {"setup": round(t1 - t0, 1), "fold_step": round(t2 - t1, 1), "threads": O.lib().lfo_num_threads()}
    d["workload"] = {"name": name, "ccs": ccs, "ring": wl.ring, "s": wl.s, "kappa": wl.kappa, "K": wl.K, "L": wl.L, "B": wl.B, "t": wl.t, "nnz": [int(len(c)) for c in wl.col]}
    print(name, ccs, d["oracle_seconds"], flush=True)
    return d


if __name__ == "__main__":
    names = sys.argv[1:] or ["C2", "T18", "C4", "B14", "C3"]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for n in names:
        base, _, ccs = n.partition("/")
        out[n] = run(base, ccs or "r1cs")
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
