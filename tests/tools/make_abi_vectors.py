"""Generates tests/golden/abi_vectors.json: input/output vectors of every C-ABI entry point at toy sizes plus one complete fold step
(T8 in full, C1 as per-section digests), computed by the CPU ORACLE ONLY (last row).  With these committed, the GPU tests in
tests/test_gpu_golden_vectors.py check the product without the oracle library.

    python tests/tools/make_abi_vectors.py

Inputs come from the indexable SplitMix64 stream of latticefold_amd/workload.py (seed, start, count are recorded, not the words), outputs
are stored in full for the element-wise entry points and as words for the protocol objects."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "tests", "golden", "abi_vectors.json")

from latticefold_amd.workload import make_workload, splitmix_fq  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def L(a):
    return [int(x) for x in np.asarray(a, dtype=np.uint64).reshape(-1)]


def ring_vectors(ring):
    if ring == "goldilocks":
        import lfo as O
    else:
        import lfo_bb as O
    RE, TAU, P = O.RE, O.TAU, O.P
    d = {"ring": ring, "RE": RE, "TAU": TAU}
    rnd = lambda seed, n: splitmix_fq(seed, 0, n * RE, ring).reshape(n, RE)
    x = rnd(11, 5)
    d["crt"] = {"seed": 11, "count": 5, "out": L(O.crt(x))}
    d["icrt"] = {"seed": 11, "count": 5, "out": L(O.icrt(x))}
    # edge coefficients for the digit rule: 0, +-1, +-B/2, +-(B/2 +- 1), (p-1)/2, (p+1)/2, p-1
    B = 1 << 8
    edge = [0, 1, P - 1, B // 2, P - B // 2, B // 2 + 1, P - B // 2 - 1, B // 2 - 1, P - B // 2 + 1, (P - 1) // 2, (P + 1) // 2, P - 1]
    e = np.zeros((2, RE), dtype=np.uint64)
    e.reshape(-1)[:len(edge)] = np.array(edge, dtype=np.uint64)
    e[1] = rnd(12, 1)[0] % np.uint64(1 << 20)
    digs = 9 if ring == "goldilocks" else 4
    d["decompose"] = {"input": L(e), "base": B, "digits": digs,
                      "layout0": L(O.decompose(e, B, digs, 0)), "layout1": L(O.decompose(e, B, digs, 1))}
    sm = (rnd(13, 6) % np.uint64(7)).astype(np.uint64)
    d["recompose"] = {"seed": 13, "count_out": 2, "digits": 3, "base": 16, "mod": 7, "out": L(O.recompose(sm, 16, 3))}
    A = splitmix_fq(14, 0, 3 * 10 * RE, ring).reshape(3, 10, RE)
    f = rnd(15, 10)
    d["ajtai_commit"] = {"seed_A": 14, "kappa": 3, "n": 10, "seed_f": 15, "out": L(O.ajtai_commit(A, 3, 10, f))}
    pt = splitmix_fq(16, 0, 4 * TAU, ring).reshape(4, TAU)
    emb = np.stack([np.tile(c, 8) for c in pt])
    eq = O.build_eq(emb)
    d["build_eq"] = {"seed": 16, "nv": 4, "out_slot0": L(eq[:, :TAU])}
    tb = rnd(17, 13)
    d["mle_eval"] = {"seed_table": 17, "len": 13, "seed_point": 16, "nv": 4, "out": L(O.mle_eval(tb, emb))}
    return d, O


def fold_vectors(name, full):
    wl = make_workload(name)
    if wl.ring == "goldilocks":
        import lfo as O
    else:
        import lfo_bb as O
    inst = O.Instance(wl)
    A = wl.ajtai_matrix()
    f = inst.witness_from_w_ccs(wl.w_ccs)
    cm = O.ajtai_commit(A, wl.kappa, wl.N, O.crt(f))
    cccs = np.concatenate([cm, wl.x_ccs])
    acc, lin = inst.linearize(O.Transcript(), cccs, f)
    lc, f0, proof = inst.fold_step(O.Transcript(), A, acc, f, cccs, f)
    lcs, dec = inst.decomposition_prove(O.Transcript(), A, acc, f)
    d = {"workload": name, "sha": {"f_coeff": sha(f), "cccs": sha(cccs), "acc": sha(acc), "lin_proof": sha(lin), "lcccs_out": sha(lc),
                                    "f0_ntt": sha(f0), "proof": sha(proof), "dec_proof_of_acc": sha(dec), "dec_lcccs_of_acc": sha(lcs)}}
    if full:
        d["cccs"] = L(cccs); d["acc"] = L(acc); d["lcccs_out"] = L(lc); d["proof"] = L(proof)
    return d


if __name__ == "__main__":
    out = {"generator": "tests/tools/make_abi_vectors.py (oracle/liblfo*.so only)"}
    for ring in ("goldilocks", "babybear"):
        out[ring], _ = ring_vectors(ring)
    out["fold_T8"] = fold_vectors("T8", True)
    out["fold_C1"] = fold_vectors("C1", False)
    out["fold_B6"] = fold_vectors("B6", False)
    json.dump(out, open(OUT, "w"), separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
