"""Generates tests/golden/chain_digests.json: SHA-256 digests of a CHAIN of fold steps (IVC style: every step ingests a NEW witness of the same constraint
system, commits it and folds it into the carried accumulator), computed by the CPU ORACLE ONLY (oracle/liblfo*.so -- no GPU, no product code beyond the
numpy workload generator).

    python tests/tools/make_chain_digests.py [name[:steps] ...]        default: T14:4 C2:4 B10:3

acc_0 = linearization of the base instance (fresh transcript).  Step j = 1..S: w_j = workload.chain_w_ccs(wl, j) -> Witness::from_w_ccs (arith.rs:230-248)
-> cm_j = Witness::commit (arith.rs:357-362) -> (acc_j, f_j, proof_j) = NIFSProver::prove(acc_{j-1}, f_{j-1}, cm_j || x, w_j) under a fresh transcript
(nifs.rs:48-103; chaining as in nifs/tests.rs:58-117).  Per step: digests of cm_j, the folded LCCCS, the folded witness (NTT form) and the proof, plus the
folded witness's max-norm.  tests/test_gpu_chain.py replays the chain through the C ABI and compares step by step.
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
OUT = os.path.join(ROOT, "tests", "golden", "chain_digests.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def run(name, steps):
    from latticefold_amd.workload import chain_w_ccs, make_workload
    wl = make_workload(name)
    if wl.ring == "goldilocks":
        import lfo as O
        p = 0xFFFFFFFF00000001
    else:
        import lfo_bb as O
        p = 15 * 2**27 + 1
    t0 = time.time()
    inst = O.Instance(wl)
    A = inst.ajtai_matrix()
    f_acc = inst.witness_from_w_ccs(wl.w_ccs)
    cm = O.ajtai_commit(A, wl.kappa, wl.N, O.crt(f_acc))
    acc, _ = inst.linearize(O.Transcript(), np.concatenate([cm, wl.x_ccs]), f_acc)
    out = {"acc0": sha(acc), "steps": []}
    for j in range(1, steps + 1):
        w = chain_w_ccs(wl, j)
        f_j = inst.witness_from_w_ccs(w)
        cm_j = O.ajtai_commit(A, wl.kappa, wl.N, O.crt(f_j))
        cccs = np.concatenate([cm_j, wl.x_ccs])
        lc, f0, proof = inst.fold_step(O.Transcript(), A, acc, f_acc, cccs, f_j)
        rc, _ = inst.verify(O.Transcript(), acc, cccs, proof)
        assert rc == 0, (name, j, rc)
        f_acc = O.icrt(f0)
        c = f_acc.astype(np.uint64)
        norm = int(np.where(c > np.uint64((p - 1) // 2), np.uint64(p) - c, c).max())
        assert norm < wl.B // 2, (name, j, norm)
        out["steps"].append({"w_ccs": sha(w), "cm": sha(cm_j), "lcccs": sha(lc), "f_ntt": sha(f0), "proof": sha(proof), "norm": norm,
                             "first_words": {"cm": [int(x) for x in np.asarray(cm_j).reshape(-1)[:3]], "proof_last": [int(x) for x in np.asarray(proof).reshape(-1, wl.RE)[-1][:3]]}})
        acc = lc
        print(name, "step", j, "norm", norm, round(time.time() - t0, 1), "s", flush=True)
    out["oracle_seconds"] = round(time.time() - t0, 1)
    out["workload"] = {"name": name, "ring": wl.ring, "s": wl.s, "kappa": wl.kappa, "steps": steps}
    return out


if __name__ == "__main__":
    items = sys.argv[1:] or ["T14:4", "C2:4", "B10:3"]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for it in items:
        n, _, st = it.partition(":")
        out[n] = run(n, int(st or 4))
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
