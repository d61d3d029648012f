#!/usr/bin/env python3
"""Times PlusProver::prove (crates/latticefold-plus/src/plus.rs:77-108) on the GPU at the reference's test shape (plus.rs:148-217: n = 2^15, kappa 2, k 2,
two fresh instances) and at larger n, with the oracle's CPU restatement beside it at the smallest size.  Prints one JSON line per shape.
usage: python tools/bench_lfplus.py [--nvars 15 16 17 18] [--rounds 3] [--cpu]"""
import argparse
import json
import os
import sys
import time
from math import ceil, log, sqrt

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from latticefold_amd import plus  # noqa: E402

D, P = 16, plus.P


def shape(nvars, kappa, k, L=3):
    a, c = 16 * 128 * L, 8 + 16 * k + 1
    est = ceil((a + sqrt(a * a + 4 * a * c)) / 2)         # utils::estimate_bound (utils.rs:102-112)
    B = est + 1 if k == 2 else est // 2                   # plus.rs:165 (test_prove) / benches/e2e.rs:71 (k = 4)
    return 1 << nvars, B, ceil(log(P) / log(8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvars", type=int, nargs="*", default=[15, 16, 17, 18])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--kappa", type=int, default=2)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--fresh", type=int, default=2, help="fresh instances folded by the one prove (benches/e2e.rs folds L = 2..5 copies)")
    ap.add_argument("--resident", action="store_true", help="witnesses preloaded, accumulator left on the device (no PCIe traffic of witnesses inside the timed call)")
    ap.add_argument("--cpu", action="store_true", help="also time the oracle (CPU restatement) at the first size")
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    for idx, nvars in enumerate(a.nvars):
        n, B, l = shape(nvars, a.kappa, a.k, max(3, a.fresh))
        A = rng.integers(0, P, size=(a.kappa, n, D), dtype=np.uint64)
        r1cs = plus.r1cs_decomposed_square((plus.identity_csr(n // a.k),) * 3, n, B, a.k)
        params = plus.PlusParameters(plus.LinParameters(a.kappa, plus.DecompParameters(8, a.k, l)), B)
        zs = []
        for _ in range(a.fresh):
            z = np.zeros((n // a.k, D), dtype=np.uint64)
            z[:, 0] = rng.integers(0, 2, size=n // a.k)
            zs.append(z)
        times = []
        for rnd in range(a.rounds + 1):               # first pass = warm-up (allocator, kernel load)
            prover = plus.PlusProver.init(A, list(r1cs), max(1, a.fresh - 2), params, plus.PoseidonTranscript())
            comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, z, 1, B, a.k) for z in zs]
            if a.resident:
                prover.device_acc = True
                prover.preload(comps)
            t0 = time.perf_counter()
            proof = prover.prove(comps)
            times.append(time.perf_counter() - t0)
            prover.close()
        ver = plus.PlusVerifier.init(A, list(r1cs), params, plus.PoseidonTranscript())
        t0 = time.perf_counter()
        ok = ver.verify(proof)
        tv = time.perf_counter() - t0
        rec = {"op": "PlusProver::prove", "ring": "frog d=16", "n": n, "kappa": a.kappa, "k": a.k, "fresh_instances": a.fresh, "B": B, "resident_io": bool(a.resident),
               "gpu_prove_ms": round(1e3 * min(times[1:]), 2), "gpu_prove_ms_all": [round(1e3 * t, 2) for t in times[1:]], "host_verify_ms": round(1e3 * tv, 2),
               "verified": bool(ok)}
        if a.cpu and idx == 0:
            import lfp
            orc = lfp.PlusOracle(A, list(r1cs), a.kappa, 8, a.k, l, B, lfp.Transcript())
            t0 = time.perf_counter()
            orc.prove([(lfp.gadget_decompose(z, B, a.k), r1cs) for z in zs])
            rec["cpu_oracle_prove_ms"] = round(1e3 * (time.perf_counter() - t0), 1)
            rec["cpu_oracle"] = "oracle/lfp*.c, 1 thread (a restatement: parity unpinned beyond the KATs, see oracle/lfp.h)"
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
