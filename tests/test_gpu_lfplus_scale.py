"""LatticeFold+ `PlusProver::prove` (crates/latticefold-plus/src/plus.rs:77-108) on the GPU at the reference's end-to-end bench shapes
(benches/utils/mod.rs:282-301: n = 65536 / 131072, L = 3, k = 4, kappa = 2) and at BASELINE configs[4]'s 2^20 rows, against the committed ORACLE-ONLY
fixtures tests/golden/lfplus_digests.json (tests/tools/make_lfplus_digests.py: SHA-256 of every proof field, the folded accumulator and the next transcript
challenge).  The small-size word-for-word comparisons against the live oracle are tests/test_gpu_lfplus_prover.py."""
import hashlib
import json
import os

import numpy as np
import pytest

from latticefold_amd import plus

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lfplus_digests.json")
CM_KEYS = ("msgs", "r", "e", "b", "v", "a", "bb", "c", "comh", "pa", "ea", "pb", "eb", "ro", "cm_g", "vo", "fcoms")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def digests(proof, acc, challenge):
    d = {"final_challenge": int(challenge), "acc_F0": _sha(acc[0]), "acc_F1": _sha(acc[1])}
    for i, lp in enumerate(proof["lproof"]):
        for key in ("msgs", "r", "evals"):
            d[f"lproof{i}_{key}"] = _sha(lp[key])
    for key in CM_KEYS:
        d[f"cm_{key}"] = _sha(proof["cmproof"][key])
    for key in ("cm_g", "ro", "vo"):
        d[f"linb2x_{key}"] = _sha(proof["linb2x"][key])
    for key in ("C0", "C1", "v0", "v1"):
        d[f"dproof_{key}"] = _sha(proof["dproof"][key])
    return d


def gold(name):
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/lfplus_digests.json missing (tests/tools/make_lfplus_digests.py)")
    g = json.load(open(GOLD))
    if name not in g:
        pytest.skip(f"no golden digest for {name}")
    return {k: v for k, v in g[name].items() if k not in ("oracle_seconds", "oracle_host", "workload", "first_words")}


def prove(wl, device=0):
    A, r1cs = wl.ajtai_matrix(), wl.r1cs()
    prover = plus.PlusProver.init(A, list(r1cs), max(1, wl.L - 2), wl.params(), plus.PoseidonTranscript(), device)
    try:
        comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, wl.z(i), 1, wl.B, wl.k) for i in range(wl.L)]
        proof = prover.prove(comps)
        return proof, prover.acc, prover.transcript.get_challenge(), A, r1cs
    finally:
        prover.close()


@pytest.mark.parametrize("name", ["P15", "P16", "P17", "P20"])
def test_plus_prover_matches_committed_oracle_digests(name):
    want = gold(name)
    wl = plus.make_plus_workload(name)
    proof, acc, ch, A, r1cs = prove(wl)
    got = digests(proof, acc, ch)
    bad = [k for k in want if got.get(k) != want[k]]
    assert not bad, f"{name}: fields differing from the oracle fixture: {bad}"
    ver = plus.PlusVerifier.init(A, list(r1cs), wl.params(), plus.PoseidonTranscript())
    assert ver.verify(proof), ver.stage
