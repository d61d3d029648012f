"""Generates tests/golden/lfplus_digests.json: SHA-256 digests of one complete `PlusProver::prove` (crates/latticefold-plus/src/plus.rs:77-108) at the
end-to-end bench shapes of the reference (benches/e2e.rs:57-100, benches/utils/mod.rs:282-301) and at BASELINE configs[4]'s 2^20 rows, computed by the
CPU ORACLE ONLY (oracle/liblfp.so through tests/lfp.py -- no GPU, no product code beyond the numpy workload generator latticefold_amd/plus.py::make_plus_workload).

    python tests/tools/make_lfplus_digests.py [names...]        default: P15 P16 P17 P20

One prove folds the L fresh instances of the workload (no accumulator yet, as in the bench).  The GPU tests (tests/test_gpu_lfplus_scale.py) recompute the
same proof through the C ABI -- unsharded and column-sharded over 2 / 4 ranks -- and compare digests field by field.  P20 needs ~45 GiB of host RAM (the
oracle materialises the monomial matrices M_f as dense ring elements) and a few minutes.
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
OUT = os.path.join(ROOT, "tests", "golden", "lfplus_digests.json")

CM_KEYS = ("msgs", "r", "e", "b", "v", "a", "bb", "c", "comh", "pa", "ea", "pb", "eb", "ro", "cm_g", "vo", "fcoms")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def sections(proof, acc, challenge):
    """one digest per proof field, so that a mismatch names the stage that produced it"""
    d = {"final_challenge": int(challenge), "acc_F0": sha(acc[0]), "acc_F1": sha(acc[1])}
    for i, lp in enumerate(proof["lproof"]):
        for key in ("msgs", "r", "evals"):
            d[f"lproof{i}_{key}"] = sha(lp[key])
    for key in CM_KEYS:
        d[f"cm_{key}"] = sha(proof["cmproof"][key])
    for key in ("cm_g", "ro", "vo"):
        d[f"linb2x_{key}"] = sha(proof["linb2x"][key])
    for key in ("C0", "C1", "v0", "v1"):
        d[f"dproof_{key}"] = sha(proof["dproof"][key])
    d["first_words"] = {"cm_g": [int(x) for x in np.asarray(proof["linb2x"]["cm_g"]).reshape(-1)[:3]], "v1": [int(x) for x in np.asarray(proof["dproof"]["v1"]).reshape(-1)[:3]]}
    return d


def run(name):
    import lfp
    from latticefold_amd import plus
    wl = plus.make_plus_workload(name)
    t0 = time.time()
    A, r1cs = wl.ajtai_matrix(), wl.r1cs()
    orc = lfp.PlusOracle(A, list(r1cs), wl.kappa, plus.D // 2, wl.k, wl.l, wl.B, lfp.Transcript())
    comps = [(lfp.gadget_decompose(wl.z(i), wl.B, wl.k), r1cs) for i in range(wl.L)]
    t1 = time.time()
    proof = orc.prove(comps)
    t2 = time.time()
    d = sections(proof, orc.acc, orc.tr.challenge())
    assert lfp.plus_verify(lfp.Transcript(), proof, wl.B) == 0, "the oracle's verifier rejects the oracle's proof"
    d["oracle_seconds"] = {"setup": round(t1 - t0, 1), "prove": round(t2 - t1, 1), "verify": round(time.time() - t2, 1)}
    d["workload"] = {"name": name, "nvars": wl.nvars, "L": wl.L, "k": wl.k, "kappa": wl.kappa, "B": wl.B, "l": wl.l}
    import platform
    d["oracle_host"] = os.environ.get("LF_ORACLE_HOST", platform.node() or "unknown host") + ", one thread"      # where oracle_seconds was measured (bench.py quotes it)
    print(name, d["oracle_seconds"], flush=True)
    return d


if __name__ == "__main__":
    names = sys.argv[1:] or ["P15", "P16", "P17", "P20"]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for n in names:
        out[n] = run(n)
        json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
