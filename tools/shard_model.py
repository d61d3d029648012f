#!/usr/bin/env python3
"""Per-rank compute time of a G-way sharded fold step, measured on ONE GPU.

    python tools/shard_model.py [--workload C4] [--steps 6] [--warmup 2] [--ranks 0] [--worlds 1,2,4,8] [--timeline]

For each G the context is rank r of G with the model transport (lf_set_sharding_model): every kernel and host stage does that rank's share of the
work, every exchange is enqueued in its lane's stream (zeros stand in for the peers' words), the schedule is the threaded two-lane one.  The GPU is
the rank's alone -- unlike G processes sharing the device (tools/gpu_shard_model.sh), where the serialised kernels and the blocking gloo round trips
of the test transport are what one measures.  The "proofs" are meaningless (the peers' partial sums are missing); only the time is read.

    t(G) = t_rank(G)                            measured here (compute + launch + host transcript + in-stream enqueue of every exchange)
         + n_small(G) * t_lat                   latency of a small all-gather over xGMI that the model transport does not pay (not measurable here)
         + gathered_bytes(G) * (G-1)/G / bw     the hand-over all-gathers of table slices

The second and third terms are printed for an assumed t_lat / bw (flags), labelled as assumptions."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--ranks", default="0", help="ranks to model per world: a list, or 'all'")
    ap.add_argument("--t-lat-us", type=float, default=25.0, help="ASSUMED latency of one small RCCL all-gather over xGMI beyond its enqueue (us)")
    ap.add_argument("--bw-gbs", type=float, default=100.0, help="ASSUMED per-rank all-gather bandwidth for the table hand-overs (GB/s)")
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--lfplus", default=None, metavar="WORKLOAD", help="model PlusProver::prove (LatticeFold+, e.g. P20) instead of the fold step: lfplus_set_sharding_model")
    args = ap.parse_args()
    if args.lfplus:
        return main_lfplus(args)
    import torch  # noqa: F401  (maps the ROCm runtime the way bench.py does)
    from latticefold_amd.shard_model import model_rank
    from latticefold_amd.workload import make_workload

    wl = make_workload(args.workload)
    out = []
    for G in [int(x) for x in args.worlds.split(",")]:
        ranks = range(G) if args.ranks == "all" else [int(r) for r in args.ranks.split(",") if int(r) < G]
        for r in ranks:
            rec = model_rank(wl, G, r, args.steps, args.warmup, 0, args.t_lat_us, args.bw_gbs, args.timeline)
            out.append(rec)
            print(json.dumps(rec), flush=True)
    base = next((o["ms_per_step"] for o in out if o["world"] == 1), None)
    for o in out:
        t = o.get("model", {}).get("t_ms", o["ms_per_step"])
        sp = f"  speed-up vs G=1 {base / t:.2f}x" if base else ""
        print(f"# G={o['world']} rank {o['rank']}: measured {o['ms_per_step']:.2f} ms/step, {o['exchanges_per_step']:.0f} exchanges, "
              f"{o['sent_bytes_per_step'] / 1e6:.1f} MB sent -> model {t:.2f} ms{sp}")


def main_lfplus(args):
    """LatticeFold+ (BASELINE configs[4]): one rank's share of a column-sharded PlusProver::prove, witnesses resident, accumulator left on the device"""
    import numpy as np
    import torch  # noqa: F401
    from latticefold_amd import plus
    from latticefold_amd.dist import column_shard
    wl = plus.make_plus_workload(args.lfplus)
    r1cs, zs = wl.r1cs(), [wl.z(i) for i in range(wl.L)]
    out = []
    for G in [int(x) for x in args.worlds.split(",")]:
        ranks = range(G) if args.ranks == "all" else [int(r) for r in args.ranks.split(",") if int(r) < G]
        for r in ranks:
            A = wl.ajtai_matrix(column_shard(wl.n, r, G) if G > 1 else None)
            best, n_ex, words = None, 0, 0
            for it in range(args.warmup + args.steps):
                prover = plus.PlusProver.init(A, list(r1cs), max(1, wl.L - 2), wl.params(), plus.PoseidonTranscript(), 0, (r, G, "model") if G > 1 else None)
                try:
                    comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, z, 1, wl.B, wl.k) for z in zs]
                    prover.device_acc = True
                    prover.preload(comps)
                    prover.ctxs[0].dist_stats(reset=True)
                    prover.ctxs[0].dist_stats_words(reset=True)
                    t0 = time.perf_counter()
                    prover.prove(comps)
                    dt = time.perf_counter() - t0
                    n_ex, words = prover.ctxs[0].dist_stats()[0], prover.ctxs[0].dist_stats_words()
                finally:
                    prover.close()
                if it >= args.warmup:
                    best = dt if best is None else min(best, dt)
            rec = {"workload": args.lfplus, "world": G, "rank": r, "ms_per_prove": best * 1e3, "exchanges_per_prove": n_ex, "sent_bytes_per_prove": words * 8}
            if G > 1:
                lat = n_ex * args.t_lat_us / 1e3
                bw = words * 8 * (G - 1) / (args.bw_gbs * 1e9) * 1e3
                rec["model"] = {"t_lat_us_ASSUMED": args.t_lat_us, "bw_gbs_ASSUMED": args.bw_gbs, "latency_ms": lat, "transfer_ms": bw, "t_ms": best * 1e3 + lat + bw}
            out.append(rec)
            print(json.dumps(rec), flush=True)
    base = next((o["ms_per_prove"] for o in out if o["world"] == 1), None)
    for o in out:
        t = o.get("model", {}).get("t_ms", o["ms_per_prove"])
        sp = f"  speed-up vs G=1 {base / t:.2f}x" if base else ""
        print(f"# LatticeFold+ {o['workload']} G={o['world']} rank {o['rank']}: measured {o['ms_per_prove']:.2f} ms/prove, {o['exchanges_per_prove']} exchanges, "
              f"{o['sent_bytes_per_prove'] / 1e6:.1f} MB sent -> model {t:.2f} ms{sp}")


if __name__ == "__main__":
    main()
