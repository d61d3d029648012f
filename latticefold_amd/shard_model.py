"""Per-rank time of a G-way sharded fold step, measured on ONE GPU with the model transport (lf_set_sharding_model).

The context is rank r of G: every kernel and host stage does that rank's share of the work, every exchange is enqueued in its lane's stream (zeros stand in for the
peers' words), the schedule is the threaded two-lane one.  The "proofs" are meaningless (the peers' partial sums are missing); only the time is read.

    t(G) = t_rank(G)                          measured (compute + launch + host transcript + in-stream enqueue of every exchange)
         + n_exchanges(G) * t_lat             latency of a small all-gather over xGMI that the model transport does not pay          (ASSUMED)
         + sent_bytes(G) * (G - 1) / bw       the hand-over all-gathers of table slices                                              (ASSUMED)

A prediction to check a multi-GPU run against -- never a measured scaling curve.  Used by bench.py (extra key `shard_model`) and tools/shard_model.py."""
import time

T_LAT_US_ASSUMED = 25.0
BW_GBS_ASSUMED = 100.0


def model_rank(wl, G, r=0, steps=4, warmup=2, device=0, t_lat_us=T_LAT_US_ASSUMED, bw_gbs=BW_GBS_ASSUMED, timeline=False):
    import numpy as np
    import torch
    from . import api
    ctx = api.Context(device, ring=wl.ring)
    try:
        if G > 1:
            ctx.set_sharding_model(r, G)
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        tr = api.PoseidonTranscript(ring=wl.ring)
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr)
        w_acc = wit
        made = []

        def step():
            nonlocal acc, w_acc
            lc, w0, _proof = api.NIFSProver.prove(ctx, acc, w_acc, cccs, wit, tr)
            made.append(w0)
            while len(made) > 2:
                made.pop(0).free()
            acc, w_acc = lc, w0

        for _ in range(warmup):
            step()
        ctx.dist_stats(reset=True)
        ctx.dist_stats_words(reset=True)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(device)
        ms = (time.perf_counter() - t0) * 1e3 / steps
        n_ex, us_tot, us_max = ctx.dist_stats()
        words = ctx.dist_stats_words()
        ph = ctx.phase_ms()
        rec = {"workload": wl.name, "world": G, "rank": r, "ms_per_step": ms, "exchanges_per_step": n_ex / steps,
               "enqueue_us_mean": us_tot / max(n_ex, 1), "enqueue_us_max": us_max, "sent_bytes_per_step": words * 8 / steps, "phases_ms": ph}
        if G > 1:
            lat = rec["exchanges_per_step"] * t_lat_us / 1e3
            bw = rec["sent_bytes_per_step"] * (G - 1) / (bw_gbs * 1e9) * 1e3
            rec["model"] = {"t_lat_us_ASSUMED": t_lat_us, "bw_gbs_ASSUMED": bw_gbs, "latency_ms": lat, "transfer_ms": bw, "t_ms": ms + lat + bw}
        if timeline:
            rec["timeline"] = ctx.timeline()
        return rec
    finally:
        ctx.close()


def model_summary(wl, worlds=(2, 4, 8), steps=4, warmup=2, device=0, base_ms=None):
    """the bench line's `shard_model` key: rank 0's modelled ms/step for each G, the exchange count, the serial host share, the predicted speed-up over base_ms"""
    out = {"what": "PREDICTION for bench.py --gpus G --parallelism shard (strong scaling of one fold step): rank 0 of G measured on one GPU with the model transport "
                   "+ ASSUMED xGMI terms; not a measured curve",
           "workload": wl.name, "t_lat_us_ASSUMED": T_LAT_US_ASSUMED, "bw_gbs_ASSUMED": BW_GBS_ASSUMED, "base_ms_per_step": base_ms, "per_world": {}}
    for G in worlds:
        rec = model_rank(wl, G, 0, steps, warmup, device)
        m = rec["model"]
        out["per_world"][str(G)] = {
            "rank_ms_measured": round(rec["ms_per_step"], 3), "exchanges_per_step": rec["exchanges_per_step"], "sent_MB_per_step": round(rec["sent_bytes_per_step"] / 1e6, 2),
            "latency_ms_ASSUMED": round(m["latency_ms"], 3), "transfer_ms_ASSUMED": round(m["transfer_ms"], 3), "t_ms_model": round(m["t_ms"], 3),
            "serial_host_transcript_ms": round(rec["phases_ms"].get("host_transcript", 0.0), 3),
            "speedup_vs_1gpu_model": round(base_ms / m["t_ms"], 3) if base_ms else None,
            "steps_per_s_model": round(1e3 / m["t_ms"], 2)}
    return out
