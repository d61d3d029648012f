#!/usr/bin/env python3
"""bench.py -- folding-prover steps/sec on MI355X (BASELINE.json metric).

One "step" = one complete `NIFSProver::prove` (crates/latticefold/src/nifs.rs:48-103): linearization of the new
instance + two decompositions + folding, including the host Poseidon transcript and all host<->device scalar
traffic, with the Ajtai matrix, the CCS and both witnesses already resident in HBM (as in the reference's
`bench_e2e_prover`, benches/utils.rs:619-680, which also sets everything up outside the timed closure and clones the
transcript per iteration).  Synthetic inputs: latticefold_amd/workload.py.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4]

N > 1: one process per GPU (torch.distributed / RCCL for rendezvous, barrier and the max-over-ranks clock); every rank
folds its own independent instance (seed = rank) -- "replicas", weak scaling; see DESIGN.md "Multi-GPU".
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline(target_wl, budget_s=25.0):
    """Time the CPU oracle (the C restatement of the reference algorithm, "port") on the host cores on a bounded sample:
    one fold step of the same parameter set at a smaller m; the fold step is linear in m, so steps/s scales by m'/m."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    if target_wl.ring == "babybear":
        import lfo_bb as lfo
    else:
        import lfo
    from latticefold_amd.workload import CONFIGS, Workload, make_workload

    lib = lfo.lib()
    threads = lib.lfo_num_threads()

    def run(s):
        name = f"_cpu{s}"
        base = CONFIGS[target_wl.name]
        CONFIGS[name] = (s, (1 << s) // base[2], base[2], base[3], base[4], base[5], base[6]) + tuple(base[7:])
        wl = make_workload(name)
        inst = lfo.Instance(wl)
        A = wl.ajtai_matrix()
        f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
        cm = lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_coeff))
        cccs = np.concatenate([cm, wl.x_ccs])
        acc, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
        t0 = time.perf_counter()
        inst.fold_step(lfo.Transcript(), A, acc, f_coeff, cccs, f_coeff)
        return time.perf_counter() - t0

    s = 10 if target_wl.ring == "goldilocks" else 6
    t = run(s)
    while s + 2 <= min(target_wl.s, 18) and t * 4.2 < budget_s:
        s += 2
        t = run(s)
    scale = (1 << target_wl.s) / (1 << s)
    return {
        "value": 1.0 / (t * scale),
        "unit": "steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle/ (C restatement, OpenMP {threads} threads) one fold step at m=2^{s} (same kappa/B/L/b/K) took {t:.2f} s; "
                  f"cost is linear in m, extrapolated x{scale:.0f} to m=2^{target_wl.s}",
        "sample_seconds": t,
    }


PMC_FILE = "r02d_pmc_{}.json"   # profiles/: HBM traffic per kernel from the PMC passes (falls back to null when absent)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("LF_WORKLOAD", "C4"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="opt-in throughput mode: S independent fold streams per GPU (S contexts driven by S host threads); every "
                         "stream does --steps steps, value counts all of them.  Default 1 = one prover, latency-honest ms_per_step")
    ap.add_argument("--parallelism", choices=["auto", "replicas", "shard"], default=os.environ.get("LF_PARALLELISM", "auto"),
                    help="N>1: 'shard' = ONE fold stream sharded over the GPUs (BASELINE configs[3]: column-sharded Ajtai commitments, "
                         "index-sharded sumcheck rounds and evaluations, RCCL exchanges; strong scaling, SURVEY 8e) -- the default ('auto') "
                         "for N>1, which also reports the replicas rate as an extra key; 'replicas' = one independent fold stream per GPU "
                         "(weak scaling) only")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
            sys.exit(2)

    import numpy as np
    import torch
    from latticefold_amd import api
    from latticefold_amd.workload import make_workload

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (there is no CPU fallback)", file=sys.stderr)
        sys.exit(3)
    # test hooks (single-GPU boxes): LF_FORCE_DEVICE pins every rank to one GPU, LF_DIST_BACKEND=gloo replaces RCCL
    if "LF_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["LF_FORCE_DEVICE"])
    backend = os.environ.get("LF_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    import threading

    def measure(shard):
        """setup (untimed: everything resident in HBM), W warm-up steps, K timed steps bracketed by barrier + synchronize; max over ranks"""
        wl = make_workload(args.workload, seed=0 if shard else rank)
        ctx = api.Context(local_rank, ring=wl.ring)
        transport = None
        if shard:
            from latticefold_amd import dist as lfd
            # RCCL communicators owned by the library (device-buffer all-gathers + modular-sum kernel); LF_DIST_BACKEND=gloo: host transport
            transport = lfd.init_sharding(ctx, rank, world, "auto")
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())  # generated on the device
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        tr0 = api.PoseidonTranscript(ring=wl.ring)
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr0)  # accumulator = linearized copy (benches/utils.rs:637-655)

        def step():
            lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr0.clone())
            w0.free()
            return proof

        # extra streams (opt-in): independent instances with their own context, witness and accumulator on the same GPU
        extra = []
        for sidx in range(1, max(1, args.streams)):
            if shard:
                raise SystemExit("--streams > 1 is a replicas-only mode")
            wl_s = make_workload(args.workload, seed=1000 * sidx + rank)
            ctx_s = api.Context(local_rank, ring=wl_s.ring)
            ctx_s.load_ccs(wl_s)
            sch_s = api.AjtaiCommitmentScheme(ctx_s, kappa=wl_s.kappa, n=wl_s.N, seed=wl_s.ajtai_seed())
            wit_s = api.Witness.from_w_ccs(ctx_s, wl_s.w_ccs)
            cccs_s = np.concatenate([wit_s.commit(sch_s), wl_s.x_ccs])
            tr_s = api.PoseidonTranscript(ring=wl_s.ring)
            acc_s, _ = api.LFLinearizationProver.prove(ctx_s, cccs_s, wit_s, tr_s)
            extra.append((ctx_s, acc_s, wit_s, cccs_s, tr_s, sch_s))

        def run_stream(st, n):
            ctx_s, acc_s, wit_s, cccs_s, tr_s, _ = st
            for _ in range(n):
                lc, w0, proof = api.NIFSProver.prove(ctx_s, acc_s, wit_s, cccs_s, wit_s, tr_s.clone())
                w0.free()

        def sync():
            ctx.synchronize()
            for st in extra:
                st[0].synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                if backend == "nccl":
                    dist.barrier(device_ids=[local_rank])
                else:
                    dist.barrier()

        for _ in range(args.warmup):
            step()
        for st in extra:
            run_stream(st, args.warmup)
        if shard:
            ctx.dist_stats(reset=True)
        sync()
        t0 = time.perf_counter()
        phases_acc, kstats = {}, []
        threads = [threading.Thread(target=run_stream, args=(st, args.steps)) for st in extra]
        for th in threads:
            th.start()
        for _ in range(args.steps):
            step()
            for k, v in ctx.phase_ms().items():
                phases_acc[k] = phases_acc.get(k, 0.0) + v
            kstats.append(ctx.kernel_stats())
        for th in threads:
            th.join()
        sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        ex = None
        if shard:
            n_ex, us_tot, us_max = ctx.dist_stats()
            ex = {"transport": transport, "exchanges_per_step": n_ex / args.steps, "mean_us": us_tot / max(n_ex, 1), "max_us": us_max,
                  "note": "host-side latency of one exchange (enqueue of ncclAllGather + modular-sum kernel when the transport is rccl; the whole blocking "
                          "round trip for the host transport)"}
        for st in extra:
            st[0].close()
        wit.free()
        ctx.close()
        return wl, elapsed, phases_acc, kstats, ex

    mode = args.parallelism
    shard = world > 1 and mode in ("auto", "shard")
    wl, elapsed, phases_acc, kstats, exch = measure(shard)
    replicas_extra = None
    if shard and mode == "auto":   # the independent-streams rate of the same GPUs, reported next to the sharded headline
        _, el_r, _, _, _ = measure(False)
        replicas_extra = {"value": world * args.steps / el_r, "unit": "steps/s", "ms_per_step": el_r / args.steps * 1e3, "scaling": "weak",
                          "parallelism": f"replicas x{world}: one independent fold stream per GPU, no data-path collective"}

    if rank == 0:
        E = 192 if wl.ring == "goldilocks" else 288
        steps_per_s = (1 if shard else world) * max(1, args.streams) * args.steps / elapsed
        alg = wl.alg_bytes()
        # dominant kernels, live HIP-event timing on the library's stream (lf_last_kernel_stats)
        fr_ms = sum(k["fold_round_ms"] for k in kstats)
        fr_n = sum(k["fold_round_launches"] for k in kstats)
        aj_ms = sum(k["ajtai_ms"] for k in kstats)
        aj_n = sum(k["ajtai_launches"] for k in kstats)
        P_F = 5 + 2 * wl.K * wl.tau
        fr_bytes = sum(P_F * (wl.m >> i) * E for i in range(wl.s)) / wl.s     # round i reads P*(N/2^(i-1))*E
        aj_bytes = (wl.kappa + wl.K - 1) * wl.N * E                              # batched commit: A once + K-1 witnesses
        kernels = {
            "k_fold_round(+round1)": {"avg_ms": fr_ms / max(fr_n, 1), "launches_per_step": fr_n / args.steps, "alg_bytes_per_launch": fr_bytes,
                                      "achieved_GBps": fr_bytes / (fr_ms / max(fr_n, 1) * 1e-3) / 1e9 if fr_ms else 0.0},
            "k_ajtai": {"avg_ms": aj_ms / max(aj_n, 1), "launches_per_step": aj_n / args.steps, "alg_bytes_per_launch": aj_bytes,
                        "achieved_GBps": aj_bytes / (aj_ms / max(aj_n, 1) * 1e-3) / 1e9 if aj_ms else 0.0},
        }
        # the batched commit is the largest single kernel (two launches per step); the twenty fold-round launches together are of the same order and are
        # listed next to it in `kernels`.  Fixed choice: the two totals are within a few per cent of each other, so a max() would flip from run to run.
        i8 = wl.b == 2 and not os.environ.get("LF_AJTAI_VALU")   # digit-plane commits on the int8 matrix cores (lf_ajtai_i8.hip), both rings
        dom = "k_ajtai"
        peak = 8000.0
        # HBM traffic of the dominant kernel from the PMC passes (collected separately, as rocprofv3 requires; see profiles/)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE.format(wl.name.lower()))))["kernels"]
            want = "k_ajtai_i8" if i8 else ("bb::" if wl.ring == "babybear" else "") + "k_ajtai"
            for name, k in pmc.items():
                base = name.split("<")[0]
                if base == want and k["fetch_bytes_max_corrected"] is not None:
                    traffic = k["fetch_bytes_max_corrected"] + k["write_bytes_max"]   # per launch like `achieved`: the largest launch (the K-1 batch)
        except Exception:
            traffic = None
        aj_t = aj_ms / max(aj_n, 1) * 1e-3
        src = "profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/gpu_pmc.sh; not re-measured in this run)" % PMC_FILE.format(wl.name.lower())
        if i8:
            # The K-1 digit-plane commitments of a decomposition as int8 GEMMs: (NL kappa) x (RD N) bytes of A times (RD N) x (RD planes) digits per launch
            # (Goldilocks: all 15 planes and 26 rows in one launch; BabyBear: plane groups of 8 + 7).  Useful MACs exclude the padding of the 16-wide MFMA
            # tiles.  Peak: dense int8 = 2 x the dense bf16 rate of /opt/skills/guides/MI355X_MICROARCH.md (2.5 PF -> 5 POP/s; its microbenchmark
            # ceiling is >= 3.94 POP/s).
            RD, NL, maxp, maxr = (24, 8, 16, 26) if wl.ring == "goldilocks" else (72, 4, 8, 16)
            launches_per_side = -(-(wl.K - 1) // maxp) * -(-wl.kappa // maxr)
            macs = NL * wl.kappa * RD * (wl.K - 1) * RD * wl.N / launches_per_side          # average per launch
            a_bytes = NL * wl.kappa * RD * wl.N / -(-wl.kappa // maxr) + RD * 4 * wl.N       # A row chunk as bytes (read once per launch) + the int32 planes
            kernels["k_ajtai"]["alg_bytes_per_launch"] = a_bytes
            kernels["k_ajtai"]["achieved_GBps"] = a_bytes / aj_t / 1e9 if aj_ms else 0.0
            kernels["k_ajtai_i8"] = kernels.pop("k_ajtai")
            dom = "k_ajtai_i8"
            tops = 2 * macs / aj_t / 1e12 if aj_ms else 0.0
            roof = {"bound": "mfma", "kernel": dom, "achieved": tops, "peak": 5000.0, "unit": "TOP/s (int8, 2 ops per MAC)", "frac": tops / 5000.0,
                    "flops_per_launch": 2 * macs, "traffic": traffic, "traffic_source": src,
                    "hbm": {"achieved_GBps": kernels[dom]["achieved_GBps"], "peak": peak, "frac": kernels[dom]["achieved_GBps"] / peak,
                            "note": "the same launch against the HBM roof: A streams once per launch as bytes (NL kappa RD N; floor about 1 ms at C4)"},
                    "note": "dominant kernel = the batched digit-plane commit, an exact int8 GEMM on v_mfma_i32_16x16x64_i8 (was k_ajtai on the integer multiplier: "
                            "7.1 ms / launch at C4, 4.0 ms at C3); the rest of the step stays integer-ALU-bound; whole-step algorithmic rate = %.1f GB/s = %.3f of the HBM peak"
                            % (alg * steps_per_s / world / 1e9, alg * steps_per_s / world / 1e9 / peak),
                    "kernels": kernels}
        else:
            # the binding roofline of this path is the integer multiplier (v_mad_*64_*32: 15.7e12 lane-ops/s measured, profiles/r01_microbench.txt):
            # multiply-accumulates of one batched commit = kappa*(K-1)*N*8 slots, 20 mads each (Toom-3 over F_{p^3}) / 81 (F_{p^9} schoolbook)
            mads_per_mac = 20 if wl.ring == "goldilocks" else 81
            aj_mads = wl.kappa * (wl.K - 1) * wl.N * 8 * mads_per_mac
            alu = {"kernel": "k_ajtai", "unit": "v_mad lane-op/s", "peak": 15.7e12, "achieved": aj_mads / aj_t if aj_ms else 0.0}
            alu["frac"] = alu["achieved"] / alu["peak"]
            roof = {"bound": "valu_int64", "bound_note": "neither HBM nor MFMA binds: the kernel is issue-bound on 64-bit integer multiply-adds (integer_alu below); "
                    "achieved / peak / frac are the HBM figures the bench contract asks for", "kernel": dom, "integer_alu": alu, "achieved": kernels[dom]["achieved_GBps"], "peak": peak, "unit": "GB/s",
                    "frac": kernels[dom]["achieved_GBps"] / peak, "traffic": traffic, "traffic_source": src,
                    "note": "integer-ALU-bound path (modular multiply from quarter-rate v_mad_*64_*32); whole-step algorithmic "
                            "rate = %.1f GB/s = %.3f of peak" % (alg * steps_per_s / world / 1e9, alg * steps_per_s / world / 1e9 / peak),
                    "kernels": kernels}
        out = {
            "metric": "folding-prover steps/sec (one NIFSProver::prove per step), " + ("GoldilocksRingNTT" if wl.ring == "goldilocks" else "BabyBearRingNTT"),
            "value": steps_per_s,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if shard else "weak",
            "vs_baseline": None,
            "dtype": "u64" if wl.ring == "goldilocks" else "u32 (31-bit Montgomery)",
            "data": "synthetic",
            "config": {"workload": f"{wl.name}: {'GoldilocksRingNTT' if wl.ring == 'goldilocks' else 'BabyBearRingNTT'} R1CS->CCS, m=N=2^{wl.s} rows, wit_len={wl.wit_len}, L={wl.L}, B=2^{wl.B.bit_length() - 1}, "
                                   f"b={wl.b}, K={wl.K}, kappa={wl.kappa}, t={wl.t}", "parallelism": (f"shard x{world}: one fold stream, witness columns / table rows sharded by the high index bits (Ajtai commits, linearization and folding sumcheck rounds, v/u/eta evaluations), RCCL all-gather + modular sum per exchange" if shard else f"replicas x{world}" + (f", {args.streams} independent streams per GPU" if args.streams > 1 else "")),
                       "alg_bytes_per_step": alg, "parity": "bit-exact vs in-repo CPU oracle; CRT/digit tables not yet confirmed against stark-rings@886a89f"},
            "roofline": roof,
            "phases_ms_per_step": {k: v / args.steps for k, v in phases_acc.items()},
        }
        if exch is not None:
            out["exchanges"] = exch
        if replicas_extra is not None:
            out["replicas"] = replicas_extra
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(wl)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU number over it
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        if backend == "nccl":
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
