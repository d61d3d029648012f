#!/usr/bin/env python3
"""Double-commitment bench (crates/latticefold-plus/benches/double_commitment.rs: RgInstance::from_f, WITNESS_SCALING / K_SCALING rows,
throughput unit = witness elements per second).  Inputs resident in HBM; timed with HIP events inside the library
(lfplus_rg_from_f_timed); the oracle (oracle/lfp.c, one core) is timed beside it.  One JSON line per configuration."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = [(32768, 2, 2), (65536, 4, 2), (131072, 4, 2), (1 << 20, 4, 2)]     # (n, k, kappa); the last one is ours (HBM-sized)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", type=int, default=0, help="run only the configuration with this n")
    args = ap.parse_args()
    from latticefold_amd import plus
    import lfp
    ctx = plus.PlusContext(0)
    for n, k, kappa in CONFIGS:
        if args.only and n != args.only:
            continue
        dp = plus.DecompParameters.for_frog(k)
        A = lfp.splitmix(1, 0, kappa * n * 16).reshape(kappa, n, 16)
        bound = 8 ** k // 2 - 1
        v = (lfp.splitmix(2, 0, n * 16) % np.uint64(2 * bound + 1)).astype(np.int64) - bound
        f = np.where(v < 0, np.uint64(plus.P) - (-v).astype(np.uint64), v.astype(np.uint64)).reshape(n, 16)
        ctx.set_matrix(A)
        ctx.set_witness(f)
        ms = ctx.time_rg_from_f(dp, args.iters)
        # algorithmic HBM bytes of one call: A twice (phase 1, phase 2), f once, D_f written, tau written + read, m_tau written
        bytes_ = 2 * kappa * n * 128 + n * 128 + k * n * 16 + 2 * n * 8 + n
        # integer work: 256 lazy adds per (k_i, row, j) + 256 64x64 multiply-accumulates per (row, j) + 32 per (row, j) in phase 2
        adds = k * kappa * n * 256 + kappa * n * 16
        macs = kappa * n * 256 + kappa * n * 16
        line = {"metric": "double_commitment_elements_per_s", "value": n / (ms * 1e-3), "unit": "witness ring elements/s", "ms_per_call": ms,
                "config": {"workload": f"RgInstance::from_f n={n} k={k} kappa={kappa} b=8 l={dp.l} (Frog ring, d=16)"}, "dtype": "u64",
                "roofline": {"bound": "valu_int64", "achieved": (adds * 3 + macs * 8) / (ms * 1e-3) / 1e12, "unit": "T lane-op/s (3 per lazy add, 8 per 64x64 mac)",
                             "hbm_GBps": bytes_ / (ms * 1e-3) / 1e9, "algorithmic_bytes": bytes_}}
        if not args.no_cpu and n <= 131072:
            t0 = time.perf_counter()
            lfp.rg_from_f(f, A, dp.b, dp.k, dp.l)
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": n / dt, "unit": "witness ring elements/s", "cores": 1, "kind": "port", "sample": "one full call of oracle/lfp.c lfp_rg_from_f"}
        print(json.dumps(line), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
