#!/usr/bin/env python3
"""bench.py -- folding-prover steps/sec on MI355X (BASELINE.json metric).

One "step" = one complete `NIFSProver::prove` (crates/latticefold/src/nifs.rs:48-103): linearization of the new
instance + two decompositions + folding, including the host Poseidon transcript and all host<->device scalar
traffic, with the Ajtai matrix, the CCS and both witnesses already resident in HBM (as in the reference's
`bench_e2e_prover`, benches/utils.rs:619-680, which also sets everything up outside the timed closure and clones the
transcript per iteration).  Synthetic inputs: latticefold_amd/workload.py.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4]

N = 1: one prover on one GPU ("scaling": "none").  N > 1: one process per GPU (torch.distributed for rendezvous, barrier and the
max-over-ranks clock); by default ONE fold stream is sharded over the N GPUs (BASELINE configs[3]: witness columns / table rows split by
the high index bits, the library's own RCCL communicators for the exchanges; "scaling": "strong") and the rate of N independent replicas
(one prover per GPU, seed = rank, no data-path collective; weak scaling) is reported next to it under "replicas"; `--parallelism replicas`
runs only those.  See DESIGN.md "Multi-GPU".  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def reference_probe():
    """SURVEY 8(d) Plan A: the reference's own Rayon path (`cargo bench --features parallel --bench e2e`, benches/env.rs:39-97) needs a Rust
    toolchain AND the network-fetched stark-rings crates (vendored or in a cargo registry cache).  Probe for both; report what is missing.
    Returns (runnable, description)."""
    import glob
    import shutil
    cargo = shutil.which("cargo")
    ref = os.environ.get("LF_REFERENCE_DIR")          # a checkout of NethermindEth/latticefold on THIS box, if the operator has one (never assumed)
    have_ref = bool(ref) and os.path.isfile(os.path.join(ref, "Cargo.toml"))
    vend = []
    for pat in ((os.path.join(ref, "vendor", "stark-rings*") if ref else ""), os.path.expanduser("~/.cargo/git/checkouts/stark-rings-*"),
                os.path.expanduser("~/.cargo/registry/src/*/stark-rings-*")):
        vend += glob.glob(pat) if pat else []
    missing = [w for w, ok in (("cargo", cargo), ("the reference checkout", have_ref), ("a vendored / cached stark-rings@886a89f", vend)) if not ok]
    return (not missing), ("cargo=%s, reference=%s, stark-rings=%s" % (cargo, ref if have_ref else None, vend[0] if vend else None)
                           + ("; missing: " + ", ".join(missing) if missing else ""))


def reference_baseline(target_wl, budget_s=120.0):
    """Plan A proper: time the reference's e2e prover bench on all host cores.  Only reached when reference_probe() found everything."""
    import re
    import subprocess
    ref = os.environ["LF_REFERENCE_DIR"]
    ring = "GOLDILOCKS" if target_wl.ring == "goldilocks" else "BABYBEAR"
    wit_len = min(target_wl.wit_len, 1 << 14)          # the reference's largest own row (benches/config.toml:156); linear in wit_len beyond
    env = dict(os.environ, **{ring: "1", "PROVER": "1", "E2E": "1", "KAPPA": str(target_wl.kappa), "WIT_LEN": str(wit_len), "L": str(target_wl.L),
                              "K": str(target_wl.K), "DURATION": "10", "CARGO_NET_OFFLINE": "true"})
    out = subprocess.run(["cargo", "bench", "--offline", "--features", "parallel", "--bench", "e2e"], cwd=os.path.join(ref, "crates", "latticefold"),
                         env=env, capture_output=True, text=True, timeout=budget_s * 10)
    m = re.search(r"time:\s+\[[^\]]*?([0-9.]+)\s*(ms|s)\s+[0-9.]+\s*(?:ms|s)\]", out.stdout)
    if out.returncode != 0 or not m:
        raise RuntimeError("cargo bench failed: " + (out.stderr or out.stdout)[-300:])
    t = float(m.group(1)) * (1e-3 if m.group(2) == "ms" else 1.0)
    scale = target_wl.wit_len / wit_len
    return {"value": 1.0 / (t * scale), "unit": "steps/s", "cores": os.cpu_count(), "kind": "reference",
            "sample": f"cargo bench --features parallel --bench e2e ({ring} PROVER E2E KAPPA={target_wl.kappa} WIT_LEN={wit_len} L={target_wl.L} K={target_wl.K}): "
                      f"median {t:.3f} s per prove; extrapolated x{scale:.0f} in wit_len", "sample_seconds": t}


def cpu_baseline(target_wl, budget_s=25.0):
    """Time the CPU oracle (the C restatement of the reference algorithm, "port") on the host cores on a bounded sample:
    one fold step of the same parameter set at a smaller m; the fold step is linear in m, so steps/s scales by m'/m."""
    runnable, probe = reference_probe()
    if runnable:
        try:
            r = reference_baseline(target_wl)
            r["plan_a_probe"] = probe
            return r
        except Exception as e:      # fall through to the C restatement, but say why
            probe += f"; Plan A attempted and failed: {e}"
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
        "plan_a_probe": "reference Rayon path (SURVEY 8d Plan A) not runnable here: " + probe,
    }


def _prof(kind, wl_name):
    """the committed profile of this kind for the workload, as named by profiles/latest.json -- the manifest tools/gpu_round.sh writes for ONE profiling round
    (tools/write_profile_manifest.py), so that pmc / kernel stats / SQ counters always belong to the same code state: (path, tag) or (None, None)"""
    try:
        man = json.load(open(os.path.join(ROOT, "profiles", "latest.json")))
        f = man["files"].get(kind, {}).get(wl_name)
        path = f and os.path.join(ROOT, "profiles", f)
        if path and os.path.exists(path):
            return path, man["tag"]
    except Exception:
        pass
    return None, None


# kernel families of the Goldilocks step, in schedule order.  `marks` = (start, end) wall-clock marks of lf_last_timeline bounding the phase on
# the caller thread (None = start of the step); phases of the first part overlap (two lanes), so their wall clocks are not additive.
PHASES = [
    ("linearization (lane 0)", (None, "linearization done"), ("k_lin_round", "k_lin_tail", "k_spmv<", "k_fix_final"), "latency: ~20 dependent launches + a host transcript hop per round"),
    ("digit-plane commits (lane 1)", (None, "lane 1 joined"), ("k_ajtai_i8", "k_sv_bits"), "int8 MFMA / LDS operand traffic (roofline above)"),
    ("decomposition evaluations (both lanes)", (None, "lane 1 joined"), ("k_dot_i8", "k_dot_pack_y", "k_spmv_t_eq", "k_recompose_crt", "k_sv_gemm<1, 0>", "k_sv_gemm<1, 0, 3, true>", "k_sv_vs_finish", "k_vs_combine", "k_eq_outer", "k_build_eq", "k_eq_pack_i8", "k_coef_eval_i8"), "HBM stream of z (int8 GEMM) + VALU"),
    ("host: right absorb + folding challenges", ("lane 1 joined", "fold challenges"), (), "host Poseidon chain (GPU idle)"),
    ("fold prepare + round 1 (int8 GEMM)", ("fold challenges", "round 1"), ("k_lincomb_z", "k_spmv_sum", "k_add_fhat_comb", "k_sv_pack_eq", "k_sv_sum", "k_sv_finish1", "k_sv_finish2", "k_sv_gemm<1, 0, 2", "k_sv_gemm<1, 0, 3, false>", "k_eq_pairsum"), "VALU (48 lazy products per column and slot in k_lincomb_z)"),
    ("fold rounds 2-3 (int8 GEMMs)", ("round 1", "round 3"), ("k_sv_gemm<2", "k_sv_gemm<4", "k_fold_round_g"), "VALU operand generation for the MFMAs"),
    ("fold rounds 4-6 (rounds 4-5 from tables over the digit codes, then a fused-fix round)", ("round 3", "round 6"),
     ("k_fold_round<true, 4", "k_fold_round<true, 6", "k_fold_round<true, 7", "k_fold_round<true, 1", "k_fold_r4tab", "k_fold_r5tab", "k_fold_mutab", "k_fold_round_lut"),
     "VALU issue: 4 lazy F_p^3 products per table pair in round 4 (no reduced product left), + 2 reduced in round 5, 6 + 4 from round 6 on"),
    ("fold rounds 7-10", ("round 6", "round 10"), ("k_fold_round<true, 0", "k_fix<", "k_reduce_rows"), "launch latency + host transcript"),
    ("fold tail rounds (persistent kernel)", ("round 10", "fold sumcheck"), ("k_fold_tail",), "host transcript round trips through the mailbox"),
    ("theta / eta", ("fold sumcheck", "theta/eta"), (), "int8 inner products (counted under evaluations) + host absorb"),
    ("rho + folded witness + folded instance", ("theta/eta", "fold done"), ("k_fold_witness", "k_crt_fwd", "k_i32_to_coef", "k_coef_to_i32"), "int32 convolutions (VALU)"),
]


def phase_report(wl_name, timelines, host_ms):
    """`roofline.phases`: per phase of the step the live wall clock (mean over the timed steps) next to what the committed profiles of the same
    command say about its kernels: kernel time (rocprofv3 --kernel-trace --stats), real HBM bytes (PMC FETCH_SIZE x2 + WRITE_SIZE), and the
    phase's own floors -- HBM at 8 TB/s for those bytes, VALU issue (SQ_ACTIVE_INST_VALU quad-cycles over 1024 SIMDs at 2.4 GHz), launches x 7 us."""
    import csv
    import re
    pmc_path, pmc_tag = _prof("pmc", wl_name)
    st_path, st_tag = _prof("stats", wl_name)
    sq_path, sq_tag = _prof("sq", wl_name)
    pmc = json.load(open(pmc_path))["kernels"] if pmc_path else {}
    sq = json.load(open(sq_path))["kernels"] if sq_path else {}
    stats = {}
    if st_path:
        for r in csv.DictReader(open(st_path)):
            k = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", r["Name"])).replace("lf::", "")
            stats[k] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    # steps covered by each profile: the commit kernel runs twice per step
    def steps_of(tbl, get):
        n = sum(get(v) for k, v in tbl.items() if k.startswith(("k_ajtai_i8<", "k_ajtai_i8s<", "k_ajtai_i8x<")))
        return max(1.0, n / 2.0)
    pmc_steps = steps_of(pmc, lambda v: v["launches"])
    st_steps = steps_of(stats, lambda v: v[0])
    sq_steps = steps_of(sq, lambda v: v["launches"])
    marks = {}
    for tl in timelines:
        for name, ms in tl:
            marks.setdefault(name, []).append(ms)
    mean = lambda name: (sum(marks[name]) / len(marks[name])) if name in marks else None
    out = []
    for name, (m0, m1), fams, binding in PHASES:
        t0 = 0.0 if m0 is None else mean(m0)
        t1 = mean(m1)
        ph = {"phase": name, "wall_ms": None if t0 is None or t1 is None else t1 - t0, "wall_marks": [m0 or "step start", m1], "binding": binding}
        sel = lambda tbl: [k for k in tbl if any(k.startswith(f) for f in fams)]
        if fams:
            ks = sel(stats)
            if ks:
                ph["kernel_ms"] = sum(stats[k][1] for k in ks) / st_steps / 1e6
                ph["launches"] = sum(stats[k][0] for k in ks) / st_steps
                ph["launch_floor_ms"] = ph["launches"] * 7e-3
            kp = sel(pmc)
            if kp:
                b = sum(pmc[k]["launches"] * ((pmc[k]["fetch_bytes_mean_corrected"] or 0) + (pmc[k]["write_bytes_mean"] or 0)) for k in kp) / pmc_steps
                ph["hbm_bytes"] = b
                ph["hbm_floor_ms"] = b / 8e12 * 1e3
            kq = sel(sq)
            if kq:
                quad = sum(sq[k]["wave_cycles"] * sq[k]["valu_active_frac"] for k in kq) / sq_steps
                ph["valu_floor_ms"] = quad * 4 / (1024 * 2.4e9) * 1e3
        out.append(ph)
    out.append({"phase": "host transcript (whole step, partly overlapped with GPU work)", "wall_ms": host_ms, "binding": "serial Poseidon chain, ~1.5 us per width-24 permutation (AVX-512 IFMA)"})
    # every wall-clock mark of the caller thread, mean over the timed steps (ms since the start of the step), in the order they occur
    order = sorted(marks, key=lambda k: sum(marks[k]) / len(marks[k]))
    out.append({"phase": "marks", "timeline_mean_ms": {k.strip(): round(sum(marks[k]) / len(marks[k]), 3) for k in order}})
    src = {"kernel_ms / launches": st_path and os.path.relpath(st_path, ROOT), "hbm_bytes": pmc_path and os.path.relpath(pmc_path, ROOT), "valu_floor_ms": sq_path and os.path.relpath(sq_path, ROOT),
           "note": "wall_ms is measured in this run (lf_last_timeline, caller thread; the first three phases run on two lanes and overlap); the other columns come from "
                   "committed rocprofv3 passes of this same command (its set-up launches -- the accumulator's linearization, one witness ingest -- are spread over "
                   "the profiled steps, so per-step launch counts of the first phases read a little high) and are not re-measured here"}
    return out, src


def lfplus_extra(world=1, rank=0, dist=None, device=0):
    """SURVEY 8(f) row 4 / BASELINE configs[4] next to the headline: LatticeFold+ PlusProver::prove (crates/latticefold-plus/src/plus.rs:77-108) on the Frog
    ring at the shape of the reference's end-to-end bench (benches/e2e.rs:57-100, benches/utils/mod.rs:282-301: L = 3 fresh instances, k = 4, kappa = 2) --
    its largest row n = 131072 (P17) and configs[4]'s 2^20 rows (P20); GPU prover + host verifier, after the timed region of the metric.  world > 1: the
    2^20-row prove column-sharded over the ranks (one GPU each, RCCL: lfplus_dist_init), every rank returning the same proof."""
    import numpy as np
    from latticefold_amd import plus
    from latticefold_amd.dist import column_shard, make_allgather
    gold = {}
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "lfplus_digests.json")))
    except Exception:
        pass
    rows = []
    names = ("P17", "P20") if world == 1 else ("P20",)
    if os.environ.get("LF_LFPLUS_WORKLOADS"):      # test hook: smaller shapes on the one-GPU test box
        names = tuple(os.environ["LF_LFPLUS_WORKLOADS"].split(","))
    for name in names:
        wl = plus.make_plus_workload(name)
        r1cs, zs = wl.r1cs(), [wl.z(i) for i in range(wl.L)]
        A = wl.ajtai_matrix(column_shard(wl.n, rank, world) if world > 1 else None)     # a rank uploads its columns only
        # iteration 0: untimed warm-up (first hipMalloc of every table, kernel load, empty scratch cache).  Then NREP warm runs of each form:
        # host I/O (`ms_host_io`): the witnesses cross PCIe inside the timed call and F0 / F1 come back to the host -- the region the reference's
        # PlusProver::prove signature implies for a first fold, and the region `cpu_oracle_ms` covers;
        # resident (`ms`, the contract's timed region): inputs resident (PlusProver.preload), the accumulator left on the device (lfplus_decompose_resident).
        NREP = 3
        best, proof, exch, host_io = None, None, None, None
        for it in range(1 + 2 * NREP):
            resident = it > NREP
            shard = None
            if world > 1:      # a fresh RCCL communicator per prover; under the gloo test hook (two ranks on one GPU) the host transport
                shard = (rank, world, _bcast_id(plus, dist, rank) if dist.get_backend() == "nccl" else make_allgather(dist.new_group()))
            prover = plus.PlusProver.init(A, list(r1cs), max(1, wl.L - 2), wl.params(), plus.PoseidonTranscript(), device, shard)
            try:
                comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, z, 1, wl.B, wl.k) for z in zs]
                if resident:
                    prover.device_acc = True
                    prover.preload(comps)
                if dist is not None:
                    dist.barrier()
                prover.ctxs[0].dist_stats(reset=True)
                t0 = time.perf_counter()
                proof = prover.prove(comps)
                dt = time.perf_counter() - t0
                exch = prover.ctxs[0].dist_stats()
            finally:
                prover.close()
            if it == 0:
                continue
            if resident:
                best = dt if best is None else min(best, dt)
            else:
                host_io = dt if host_io is None else min(host_io, dt)
        if dist is not None:      # the slowest rank's best time
            import torch
            t = torch.tensor([best], dtype=torch.float64)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            best = float(t.item())
        rec = {"workload": f"{name}: n = 2^{wl.nvars}, L = {wl.L} fresh instances, k = {wl.k}, kappa = {wl.kappa}, B = {wl.B}", "ms": 1e3 * best,
               "ms_host_io": 1e3 * host_io,
               "io": f"one untimed warm-up, then the min of {NREP} warm runs of each form.  ms: witnesses resident before the call, (F0, F1) left on the device; "
                     "ms_host_io: L witnesses uploaded and F0, F1 downloaded inside the call"}
        if rank == 0:
            tv, ok = None, True
            Av = A if world == 1 else np.zeros((wl.kappa, wl.n, 16), dtype=np.uint64)      # (the verifier reads the matrix's shape only)
            for _ in range(2):
                t0 = time.perf_counter()
                ok = plus.PlusVerifier.init(Av, list(r1cs), wl.params(), plus.PoseidonTranscript()).verify(proof) and ok
                dt = time.perf_counter() - t0
                tv = dt if tv is None else min(tv, dt)
            rec.update(host_verify_ms=1e3 * tv, verified=bool(ok))
            if name in gold:      # the committed oracle-only fixture of this very workload: one digest is enough to tie the timed proof to it
                import hashlib
                sha = hashlib.sha256(np.ascontiguousarray(proof["linb2x"]["cm_g"], dtype=np.uint64).tobytes()).hexdigest()
                rec["matches_oracle_fixture"] = sha == gold[name].get("linb2x_cm_g")
                rec["cpu_oracle_ms"] = 1e3 * gold[name]["oracle_seconds"]["prove"]
                rec["cpu_oracle_source"] = ("tests/golden/lfplus_digests.json: oracle/lfp*.c, one thread, timed on the host that made the fixture (oracle_host in the fixture; "
                                            "P15-P17 and P20 were made on different hosts) -- not re-measured here")
                rec["cpu_oracle_region"] = "the whole PlusProver::prove from host witnesses to host (F0, F1): compare with ms_host_io, not with ms"
        if world > 1:
            rec["parallelism"] = f"shard x{world}: double commitments, sumcheck tables and evaluations over the ranks' rows, RCCL all-gather + modular sum"
            rec["exchanges"] = {"count": exch[0], "total_us": exch[1], "max_us": exch[2]}
        rows.append(rec)
    held = plus.scratch_bytes(device)
    plus.scratch_trim(device)          # the provers are gone: give their idle scratch back to the driver before the caller returns to the main path
    return {"op": "PlusProver::prove", "ring": "Frog Z_p[X]/(X^16+1), coefficient form", "runs": rows, "scratch_cache_bytes_released": held,
            "parity": "bit-exact vs the in-repo oracle (tests/test_gpu_lfplus_scale.py, tests/test_dist_shard_lfplus.py against committed oracle-only digests); the oracle "
                      "is pinned to the reference through the transcript KATs only"}


def ajtai_extra(device=0):
    """The reference's own Ajtai bench rows (crates/latticefold/benches/ajtai.rs:15-31 "CommitNTT", build.rs:454-465): one AjtaiCommitmentScheme::commit_ntt
    of a full-range vector, batch 1, at benches/config.toml:715 (GoldilocksRingNTT, kappa 20, n 2^20) and :670 (BabyBearRingNTT, kappa 15, n 2^20).  (Witness::commit
    of the bench workload's own resident witness is timed next to the fold steps and appended to the same key.)  ms = HIP events around the whole device side of one commitment
    (digit pass, int8 contraction k_ajtai_i8g, recombination, CRT of the kappa outputs); inputs resident (the upload of f is outside), matrix generated on the device.
    frac = SURVEY 8(d) bytes (kappa + 1) N E over that time and the 8 TB/s HBM peak."""
    import numpy as np
    from latticefold_amd import api
    rows = []
    for ring, kappa, lg, ref in (("goldilocks", 20, 20, "benches/config.toml:715"), ("babybear", 15, 20, "benches/config.toml:670")):
        rec = {"op": "AjtaiCommitmentScheme::commit_ntt", "ring": ring, "kappa": kappa, "n": 1 << lg, "reference_row": ref}
        try:
            ctx = api.Context(device, ring=ring)
            n, E = 1 << lg, (192 if ring == "goldilocks" else 288)
            p = 0xFFFFFFFF00000001 if ring == "goldilocks" else 15 * 2**27 + 1
            sch = api.AjtaiCommitmentScheme(ctx, kappa=kappa, n=n, seed=0xA17A1)
            f = np.random.default_rng(7).integers(0, p, size=(n, ctx.RE), dtype=np.uint64)
            ms = []
            for _ in range(4):
                sch.commit_ntt(f)
                ms.append(ctx.kernel_stats()["ajtai_ms"])
            alg = (kappa + 1) * n * E
            rec.update({"ms": min(ms[1:]), "ms_runs": ms[1:], "alg_bytes": alg, "achieved_GBps": alg / (min(ms[1:]) * 1e-3) / 1e9, "frac": alg / (min(ms[1:]) * 1e-3) / 8e12,
                        "kernel": "k_ajtai_i8g (lf_ajtai_i8g.hip): exact int8 contraction of the byte planes of A with 10 (Goldilocks) / 5 (BabyBear) balanced base-128 digit planes of f"})
            ctx.close()
        except Exception as e:   # a reported extra: never lose the headline over it
            rec["ms"] = None
            rec["note"] = f"failed: {e!r}"
        rows.append(rec)
    return rows


def ivc_extra(name, steps, device=0, headline_ms=None, overlap=True):
    """A real chain (crates/latticefold/examples/e2e.rs, nifs/tests.rs:58-117 made a loop): every step takes a NEW witness of the workload's constraint system from
    host memory (workload.chain_w_ccs), ingests it (Witness::from_w_ccs, arith.rs:230-248: upload, ICRT, gadget decomposition), commits it (Witness::commit,
    arith.rs:357-362: the int8 general commit), folds it into the carried accumulator (NIFSProver::prove) and frees what the step replaced.  Wall clock over `steps`
    steps after two warm-up steps, bracketed by device synchronisation; parts = host wall time of the three calls.  The witnesses are generated before the loop."""
    import hashlib
    import numpy as np
    import torch
    from latticefold_amd import api
    from latticefold_amd.workload import chain_w_ccs, make_workload
    rec = {"op": "chain: from_w_ccs + Witness::commit + NIFSProver::prove per step", "workload": name, "steps": steps,
           "ingestion": "overlapped: witness j+1 uploaded / decomposed on the context's lowest-priority stream while step j folds (lf_witness_from_w_ccs_begin)" if overlap
                        else "blocking: Witness::from_w_ccs between the steps"}
    try:
        wl = make_workload(name)
        ctx = api.Context(device, ring=wl.ring)
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        tr = lambda: api.PoseidonTranscript(ring=wl.ring)
        warm = 2
        ws = [np.ascontiguousarray(chain_w_ccs(wl, j)) for j in range(1, warm + steps + 1)]
        w_acc = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        acc, _ = api.LFLinearizationProver.prove(ctx, np.concatenate([w_acc.commit(scheme), wl.x_ccs]), w_acc, tr())
        parts = {"ingest": 0.0, "commit": 0.0, "fold": 0.0}
        norm_max, last = 0, None
        t_start = None
        pending = api.Witness.from_w_ccs_begin(ctx, ws[0]) if overlap else None
        for j, w in enumerate(ws):
            if j == warm:
                torch.cuda.synchronize(device)
                parts = {k: 0.0 for k in parts}
                t_start = time.perf_counter()
            t0 = time.perf_counter()
            w_j = pending.result() if overlap else api.Witness.from_w_ccs(ctx, w)      # overlap: "ingest" = what is left to wait for
            t1 = time.perf_counter()
            cccs = np.concatenate([w_j.commit(scheme), wl.x_ccs])
            t2 = time.perf_counter()
            if overlap and j + 1 < len(ws):
                pending = api.Witness.from_w_ccs_begin(ctx, ws[j + 1])                    # the next witness crosses PCIe and is decomposed while this step folds
            lc, w_next, proof = api.NIFSProver.prove(ctx, acc, w_acc, cccs, w_j, tr())
            t3 = time.perf_counter()
            parts["ingest"] += t1 - t0; parts["commit"] += t2 - t1; parts["fold"] += t3 - t2
            w_j.free(); w_acc.free()
            acc, w_acc, last = lc, w_next, proof
        torch.cuda.synchronize(device)
        elapsed = time.perf_counter() - t_start
        ok, mx = ctx.linf_check(w_acc.f, wl.B // 2)     # (after the timed region)
        rec.update({"ms_per_step": elapsed / steps * 1e3, "steps_per_s": steps / elapsed, "parts_ms_per_step": {k: v / steps * 1e3 for k, v in parts.items()},
                    "witness_upload_bytes_per_step": int(ws[0].nbytes), "folded_norm_below_B_half": bool(ok), "folded_norm": int(mx),
                    "headline_ms_per_step": headline_ms, "over_headline_ms": (elapsed / steps * 1e3 - headline_ms) if headline_ms else None})
        try:   # tie the chain to the oracle-only fixture where one exists for this workload (tests/golden/chain_digests.json): replay its steps from a fresh accumulator
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "chain_digests.json"))).get(name)
            if gold:
                w_a = api.Witness.from_w_ccs(ctx, wl.w_ccs)
                a, _ = api.LFLinearizationProver.prove(ctx, np.concatenate([w_a.commit(scheme), wl.x_ccs]), w_a, tr())
                good = True
                for j, g in enumerate(gold["steps"], start=1):
                    w_j = api.Witness.from_w_ccs(ctx, chain_w_ccs(wl, j))
                    lc, w_n, proof = api.NIFSProver.prove(ctx, a, w_a, np.concatenate([w_j.commit(scheme), wl.x_ccs]), w_j, tr())
                    good = good and hashlib.sha256(np.ascontiguousarray(proof, dtype=np.uint64).tobytes()).hexdigest() == g["proof"]
                    w_j.free(); w_a.free()
                    a, w_a = lc, w_n
                rec["matches_oracle_chain_fixture"] = bool(good)
                rec["fixture"] = f"tests/golden/chain_digests.json[{name}]: {len(gold['steps'])} steps, oracle only (tests/tools/make_chain_digests.py)"
                w_a.free()
            else:
                rec["matches_oracle_chain_fixture"] = None
                rec["fixture"] = "none at this size (tests/test_gpu_chain.py checks C4 chains through the oracle's verifier, the commitment opening and the norm)"
        except Exception as e:
            rec["matches_oracle_chain_fixture"] = None
            rec["fixture"] = f"not checked: {e!r}"
        w_acc.free()
        ctx.close()
    except Exception as e:   # a reported extra: never lose the headline over it
        rec["ms_per_step"] = None
        rec["note"] = f"failed: {e!r}"
    return rec


def _headline_fallback(args, world, wl, elapsed, shard, lfplus):
    """the metric line without the reporting extras (used only when the sharded LatticeFold+ extra hangs: the process group is unusable afterwards)"""
    sps = (1 if shard else world) * args.steps / elapsed
    return {"metric": "folding-prover steps/sec (one NIFSProver::prove per step), " + ("GoldilocksRingNTT" if wl.ring == "goldilocks" else "BabyBearRingNTT"), "value": sps,
            "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "u64" if wl.ring == "goldilocks" else "u32 (31-bit Montgomery)", "data": "synthetic",
            "config": {"workload": wl.name, "parallelism": ("shard" if shard else "replicas") + f" x{world}"}, "roofline": None, "cpu_baseline": None, "lfplus": lfplus,
            "note": "reporting extras dropped: the sharded LatticeFold+ extra did not complete"}


def _bcast_id(plus, dist, rank):
    ids = [plus.dist_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    return ids[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("LF_WORKLOAD", "C4"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ccs", choices=["r1cs", "multi4", "multi16", "deg3"], default="r1cs",
                    help="constraint-system shape (workload.make_workload): the reference bench's 1-nnz R1CS (default, the metric's configuration); multi4 / multi16 = 4 / 16 entries "
                         "per row at pseudo-random columns, ring-valued entries in C (general CSR SpMV, arith/utils.rs:52-65); deg3 = the reference's degree-three CCS "
                         "(arith/ccs.rs:14-43: t = 4, S = {{0,1,2},{3}})")
    ap.add_argument("--no-ajtai", action="store_true", help="skip the reference's Ajtai bench rows (commit_ntt at benches/config.toml:715 / :670; an extra key, not part of the metric)")
    ap.add_argument("--chain", type=int, default=8, help="steps of the chained-folding extra key `ivc` (every step ingests and commits a new witness and folds it into the "
                                                         "carried accumulator; run for the bench workload and for C2); 0 = skip")
    ap.add_argument("--no-shard-model", action="store_true", help="skip the extra key `shard_model` (N=1: the predicted per-rank ms of --gpus 2/4/8 --parallelism shard, "
                                                                  "latticefold_amd/shard_model.py: rank 0 measured with the model transport + assumed xGMI terms)")
    ap.add_argument("--no-lfplus", action="store_true", help="skip the LatticeFold+ PlusProver::prove timing (an extra key, not part of the metric)")
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
        wl = make_workload(args.workload, seed=0 if shard else rank, ccs=args.ccs)
        ctx = api.Context(local_rank, ring=wl.ring)
        transport = None
        if shard:
            from latticefold_amd import dist as lfd
            # RCCL communicators owned by the library (device-buffer all-gathers + modular-sum kernel); LF_DIST_BACKEND=gloo: host transport
            transport = lfd.init_sharding(ctx, rank, world, "auto")
        ctx.load_ccs(wl)
        # generated on the device (the context keeps it as int8 byte planes only: lfhip.h)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        # accumulator = linearized copy (benches/utils.rs:637-655).  Both calls start from a FRESH transcript, exactly as tests/tools/make_scale_digests.py drives the
        # oracle: the proof of every timed step is then the proof whose digests are committed in tests/golden/scale_digests.json (checked after the timed loop)
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, api.PoseidonTranscript(ring=wl.ring))
        tr0 = api.PoseidonTranscript(ring=wl.ring)
        last = {}

        def step():
            lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr0.clone())
            w0.free()
            last["lc"], last["proof"] = lc, proof
            return proof

        # extra streams (opt-in): independent instances with their own context, witness and accumulator on the same GPU
        extra = []
        for sidx in range(1, max(1, args.streams)):
            if shard:
                raise SystemExit("--streams > 1 is a replicas-only mode")
            wl_s = make_workload(args.workload, seed=1000 * sidx + rank, ccs=args.ccs)
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
        phases_acc, kstats, timelines = {}, [], []
        threads = [threading.Thread(target=run_stream, args=(st, args.steps)) for st in extra]
        for th in threads:
            th.start()
        for _ in range(args.steps):
            step()
            for k, v in ctx.phase_ms().items():
                phases_acc[k] = phases_acc.get(k, 0.0) + v
            kstats.append(ctx.kernel_stats())
            timelines.append(ctx.timeline())
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
        # the proof of the LAST TIMED step against the committed oracle-only fixture of this workload (seed 0: rank 0 of a replica run, every rank of a sharded one)
        fixture_info.clear()
        if (shard or rank == 0) and last:
            try:
                import hashlib
                fkey = wl.name if args.ccs == "r1cs" else f"{wl.name}/{args.ccs}"
                gold = json.load(open(os.path.join(ROOT, "tests", "golden", "scale_digests.json"))).get(fkey)
                if gold:
                    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()
                    fixture_info.update(matches_oracle_fixture=bool(sha(last["proof"]) == gold.get("proof") and sha(last["lc"]) == gold.get("lcccs_out")),
                                        fixture="tests/golden/scale_digests.json[%s]: sha256 of the whole proof and of the folded LCCCS of the last timed step "
                                                "(oracle-only fixture, tests/tools/make_scale_digests.py)" % fkey)
            except Exception as e:      # (a missing fixture is not a bench failure)
                fixture_info.update(matches_oracle_fixture=None, fixture=f"not checked: {e!r}")
        free_b, total_b = ctx.device_memory()
        mem_info["hbm_in_use_gib"] = (total_b - free_b) / 2.0 ** 30   # whole device, this process being its only user: context, witnesses, torch's own few MB
        if world == 1 and not shard and not args.no_ajtai:
            # Witness::commit (arith.rs:357-362) of the bench's own resident witness: the cm_i every fresh instance of a real chain needs.  After the timed loop
            # (a reported extra); device side of the call (HIP events: digit pass of the int32 planes, int8 contraction, recombination, CRT)
            try:
                wms = []
                for _ in range(4):
                    wit.commit(scheme)
                    wms.append(ctx.kernel_stats()["ajtai_ms"])
                alg_w = (wl.kappa + 1) * wl.N * (192 if wl.ring == "goldilocks" else 288)
                mem_info["witness_commit"] = {"op": "Witness::commit", "workload": wl.name, "kappa": wl.kappa, "n": wl.N, "ms": min(wms[1:]), "ms_runs": wms[1:], "alg_bytes": alg_w,
                                              "achieved_GBps": alg_w / (min(wms[1:]) * 1e-3) / 1e9, "frac": alg_w / (min(wms[1:]) * 1e-3) / 8e12,
                                              "kernel": "k_ajtai_i8g (5 balanced base-128 digit planes of the handle's int32 coefficients)"}
            except Exception as e:
                mem_info["witness_commit"] = {"op": "Witness::commit", "ms": None, "note": f"failed: {e!r}"}
        for st in extra:
            st[0].close()
        wit.free()
        ctx.close()
        return wl, elapsed, phases_acc, kstats, ex, timelines

    mode = args.parallelism
    mem_info, fixture_info = {}, {}
    shard = world > 1 and mode in ("auto", "shard")
    replicas_extra, shard_note = None, None
    if shard:
        # N > 1, default: the sharded step (BASELINE configs[3]) is the headline.  It has never run on more than one GPU, so it must not be able to cost the
        # run its number: the independent replicas (no data-path collective) are measured FIRST; the sharded measurement then runs under a watchdog, the ranks agree
        # on its outcome through the rendezvous store (not through the communicator that may be the broken part), and on any failure or time-out every rank falls back
        # to the replicas line ("scaling": "weak", the reason in `note`) and leaves without touching the process group again.
        res_r = measure(False) if mode == "auto" else None
        if res_r is not None:
            replicas_extra = {"value": world * args.steps / res_r[1], "unit": "steps/s", "ms_per_step": res_r[1] / args.steps * 1e3, "scaling": "weak",
                              "parallelism": f"replicas x{world}: one independent fold stream per GPU, no data-path collective"}
        box = {}

        def _run_shard():
            try:
                torch.cuda.set_device(local_rank)
                if os.environ.get("LF_BENCH_FORCE_SHARD_FAIL") == str(rank):      # (test hook: this rank's sharded measurement fails)
                    raise RuntimeError("forced failure of the sharded measurement (LF_BENCH_FORCE_SHARD_FAIL)")
                box["r"] = measure(True)
            except BaseException as e:      # noqa: BLE001 -- anything: the fallback decides
                box["e"] = repr(e)
        th = threading.Thread(target=_run_shard, daemon=True)
        th.start()
        th.join(float(os.environ.get("LF_SHARD_TIMEOUT", "420")))
        ok_here = (not th.is_alive()) and "r" in box
        all_ok = ok_here
        try:      # agreement over the rendezvous store -- through a client connection of its own: the process group's client may be blocked inside the stuck measurement
            from datetime import timedelta
            store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world_size=world, is_master=False, wait_for_workers=False,
                                  timeout=timedelta(seconds=float(os.environ.get("LF_SHARD_AGREE_TIMEOUT", "120"))))
            store.set(f"lf_shard_ok_{rank}", "1" if ok_here else "0")
            keys = [f"lf_shard_ok_{r}" for r in range(world)]
            store.wait(keys, timedelta(seconds=float(os.environ.get("LF_SHARD_AGREE_TIMEOUT", "120"))))
            all_ok = all(store.get(k) == b"1" for k in keys)
        except Exception as e:      # a peer never reported: it is stuck or dead
            all_ok = False
            box.setdefault("e", f"no agreement with the peers: {e!r}")
        if all_ok:
            wl, elapsed, phases_acc, kstats, exch, timelines = box["r"]
        else:
            shard_note = "sharded step not measured (" + (box.get("e") or ("timed out" if th.is_alive() else "a peer rank failed")) + "): the headline is the replicas rate"
            if res_r is None:
                if rank == 0:
                    print("bench.py: " + shard_note + "; --parallelism shard has no fallback", file=sys.stderr)
                os._exit(4)
            if rank == 0:
                line = _headline_fallback(args, world, res_r[0], res_r[1], False, None)
                line["note"] = shard_note
                print(json.dumps(line), flush=True)
            os._exit(0)      # the communicator may be unusable: no further collective, no destroy_process_group
    else:
        wl, elapsed, phases_acc, kstats, exch, timelines = measure(False)

    # BASELINE configs[4]: the 2^20-row LatticeFold+ prove sharded over the same GPUs (an extra key; every rank takes part).  A watchdog bounds it: an exchange
    # that never completes must not cost the headline -- the ranks then drop the key and go on.
    lfplus_sharded = None
    if world > 1 and not args.no_lfplus:
        box = {}

        def _run():
            try:
                box["r"] = lfplus_extra(world, rank, dist, local_rank)
            except Exception as e:
                box["r"] = {"op": "PlusProver::prove", "runs": None, "note": f"failed: {e!r}"}
        th = threading.Thread(target=_run, daemon=True)
        th.start()
        th.join(240.0)
        lfplus_sharded = box.get("r", {"op": "PlusProver::prove", "runs": None, "note": "timed out (watchdog): dropped"})
        if th.is_alive():      # a stuck collective: nothing after this point may touch the process group again
            if rank == 0:
                print(json.dumps(_headline_fallback(args, world, wl, elapsed, shard, lfplus_sharded)), flush=True)
            os._exit(0)

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
        i8 = wl.b == 2   # digit-plane commits on the int8 matrix cores (lf_ajtai_i8.hip), both rings
        dom = "k_ajtai"
        peak = 8000.0
        # HBM traffic of the dominant kernel from the PMC passes (collected separately, as rocprofv3 requires; see profiles/)
        traffic = None
        try:
            pmc_path, _ = _prof("pmc", wl.name.lower())
            pmc = json.load(open(pmc_path))["kernels"]
            want = "k_ajtai_i8" if i8 else ("bb::" if wl.ring == "babybear" else "") + "k_ajtai"
            for name, k in pmc.items():
                base = name.split("<")[0]
                if (base == want or (i8 and base in (want + "s", want + "x"))) and k["fetch_bytes_max_corrected"] is not None:   # (k_ajtai_i8s / _i8x: the specialised-wave kernels of the 24- / 72-ring)
                    traffic = k["fetch_bytes_max_corrected"] + k["write_bytes_max"]   # per launch like `achieved`: the largest launch (the K-1 batch)
        except Exception:
            traffic = None
        aj_t = aj_ms / max(aj_n, 1) * 1e-3
        src = "%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, tools/gpu_pmc.sh; not re-measured in this run)" % (
            os.path.relpath(_prof("pmc", wl.name.lower())[0], ROOT) if _prof("pmc", wl.name.lower())[0] else "profiles/: no PMC file")
        if i8:
            # The K-1 digit-plane commitments of a decomposition as int8 GEMMs: (NL kappa) x (RD N) bytes of A times (RD N) x (RD planes) digits per launch
            # (Goldilocks: all 15 planes and 26 rows in one launch; BabyBear: plane groups of 8 + 7).  Useful MACs exclude the padding of the 16-wide MFMA
            # tiles.  Peak: dense int8 = 2 x the dense bf16 rate of /opt/skills/guides/MI355X_MICROARCH.md (2.5 PF -> 5 POP/s; its microbenchmark
            # ceiling is >= 3.94 POP/s).
            RD, NL, maxp, maxr = (24, 8, 16, 26) if wl.ring == "goldilocks" else (72, 4, 8, 16)
            row_chunks = -(-wl.kappa // maxr)
            lps = aj_n / args.steps if aj_n else 2.0                                         # launches per step (measured): 1 per row chunk and plane group
            sides_per_launch = 2.0 * -(-(wl.K - 1) // maxp) * row_chunks / lps               # 2 = both decompositions' planes in one launch (paired workgroups)
            macs = 2 * NL * wl.kappa * RD * (wl.K - 1) * RD * wl.N / lps                     # average per launch
            a_bytes = NL * wl.kappa * RD * wl.N / row_chunks + sides_per_launch * RD * 4 * wl.N   # A row chunk as bytes (from HBM once per launch) + the int32 planes
            # SURVEY 8(d)'s own figure for the same launch -- element bytes E, not byte planes: the rows of A this launch streams + the digit-plane
            # witnesses it commits, (kappa_launch + planes_launch) N E  (= (kappa + K - 1) N E when one launch covers a whole decomposition)
            bytes_8d = (wl.kappa / row_chunks + 2 * (wl.K - 1) * row_chunks / lps) * wl.N * E
            kernels["k_ajtai"]["alg_bytes_8d_per_launch"] = bytes_8d
            kernels["k_ajtai"]["alg_bytes_per_launch"] = a_bytes
            kernels["k_ajtai"]["achieved_GBps"] = a_bytes / aj_t / 1e9 if aj_ms else 0.0
            # the instantiation that is launched (lf_ajtai_i8.hip launch_ajtai_i8): the specialised-wave kernels for the 13-row-tile shape of the 24-ring and the
            # 4-row-tile shape of the 72-ring, the generic one otherwise
            mt = -(-NL * -(-wl.kappa // row_chunks) // 16)
            dom = "k_ajtai_i8s<false, true>" if (RD, mt) == (24, 13) else ("k_ajtai_i8x<72, 4, false>" if (RD, mt) == (72, 4) else "k_ajtai_i8")   # (as rocprofv3 prints them)
            kernels[dom] = kernels.pop("k_ajtai")
            tops = 2 * macs / aj_t / 1e12 if aj_ms else 0.0
            gbps_8d = bytes_8d / aj_t / 1e9 if aj_ms else 0.0
            roof = {"bound": "hbm", "kernel": dom, "achieved": gbps_8d, "peak": peak, "unit": "GB/s", "frac": gbps_8d / peak, "alg_bytes_per_launch": bytes_8d,
                    "frac_note": "THE CONTRACT'S NUMBER: SURVEY 8(d) algorithmic bytes of the batched commit, (kappa + planes) N E per launch, over the live HIP-event "
                                 "duration of the launch and the 8 TB/s HBM peak",
                    "traffic": traffic, "traffic_source": src,
                    "mfma": {"achieved": tops, "peak": 5000.0, "unit": "TOP/s (int8, 2 ops per MAC)", "frac": tops / 5000.0, "flops_per_launch": 2 * macs,
                             "note": "the kernel is an exact int8 GEMM on v_mfma_i32_16x16x64_i8; peak = 2 x the dense bf16 rate of the guide (it lists no int8 figure; "
                                     "its microbenchmark ceiling is 3.94 POP/s)"},
                    "hbm_8d": {"achieved_GBps": gbps_8d, "peak": peak, "frac": gbps_8d / peak, "alg_bytes_per_launch": bytes_8d, "note": "= achieved / peak / frac above (kept for readers of earlier rounds' lines)"},
                    "hbm": {"achieved_GBps": kernels[dom]["achieved_GBps"], "peak": peak, "frac": kernels[dom]["achieved_GBps"] / peak,
                            "note": "the same launch with the bytes this implementation really has to move: A once per launch as byte planes (NL kappa RD N) + the int32 witness planes"},
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
            "scaling": "strong" if shard else ("none" if world == 1 else "weak"),
            "vs_baseline": None,
            "dtype": "u64" if wl.ring == "goldilocks" else "u32 (31-bit Montgomery)",
            "data": "synthetic",
            "config": {"workload": f"{wl.name}: {'GoldilocksRingNTT' if wl.ring == 'goldilocks' else 'BabyBearRingNTT'} R1CS->CCS, m=N=2^{wl.s} rows, wit_len={wl.wit_len}, L={wl.L}, B=2^{wl.B.bit_length() - 1}, "
                                   f"b={wl.b}, K={wl.K}, kappa={wl.kappa}, t={wl.t}" + ("" if args.ccs == "r1cs" else f", constraint system '{args.ccs}' (nnz per matrix {[int(len(c)) for c in wl.col]}, degree {wl.d}) -- NOT the metric's configuration"), "parallelism": (f"shard x{world}: one fold stream, witness columns / table rows sharded by the high index bits (Ajtai commits, linearization and folding sumcheck rounds, v/u/eta evaluations), RCCL all-gather + modular sum per exchange" if shard else f"replicas x{world}" + (f", {args.streams} independent streams per GPU" if args.streams > 1 else "")),
                       "alg_bytes_per_step": alg, "hbm_in_use_gib": round(mem_info.get("hbm_in_use_gib", 0.0), 2),
                       "folded_witness": ("Witness::from_f in full inside the timed step (arith.rs:299-313): the folded witness leaves the step as int32 coefficient planes (f_coeff, what the next step reads), "
                                          "f_0 in NTT form and w_ccs, all three on the device (lf_witness_get_f / _get_w_ccs only download)"),
                       "parity": "bit-exact vs in-repo CPU oracle; CRT/digit tables not yet confirmed against stark-rings@886a89f"},
            "roofline": roof,
            "phases_ms_per_step": {k: v / args.steps for k, v in phases_acc.items()},
        }
        out["config"].update(fixture_info)
        if wl.ring == "goldilocks" and not shard and args.streams == 1:
            try:
                ph, ph_src = phase_report(wl.name.lower(), timelines, phases_acc.get("host_transcript", 0.0) / args.steps)
                roof["phases"] = ph
                roof["phases_source"] = ph_src
            except Exception as e:   # a reporting extra: never lose the headline over it
                roof["phases"] = None
                roof["phases_source"] = f"failed: {e!r}"
        try:   # the whole step against the HBM peak with the bytes it REALLY moves: sum over the committed PMC passes (FETCH x 2 + WRITE per launch x launches per step)
            pmc_path, _ = _prof("pmc", wl.name.lower())
            st_path, _ = _prof("stats", wl.name.lower())
            if pmc_path and st_path and not shard and args.streams == 1:
                import csv
                pmc = json.load(open(pmc_path))["kernels"]
                calls = {}
                for r in csv.DictReader(open(st_path)):
                    calls[r["Name"].replace("void ", "").replace("lf::", "").replace("bb::", "").split("(")[0]] = int(r["Calls"])
                setup = ("k_ajtai_icrt_pack_i8", "k_ajtai_unpack_i8", "k_fill_ajtai", "k_ajtai<", "k_aos_to_soa", "k_decompose", "k_coef_to_i32")   # matrix install / witness ingest: not per step
                nsteps = 7.0   # tools/gpu_round.sh profiles 5 timed + 2 warm-up steps
                tot = sum((v["fetch_bytes_mean_corrected"] + v["write_bytes_mean"]) * calls[k] / nsteps for k, v in pmc.items()
                          if k in calls and not k.startswith(setup) and v["fetch_bytes_mean_corrected"] is not None)
                roof["whole_step_traffic"] = {"hbm_bytes_per_step": tot, "achieved_GBps": tot / (elapsed / args.steps) / 1e9, "frac": tot / (elapsed / args.steps) / 1e9 / peak,
                                              "source": f"{os.path.relpath(pmc_path, ROOT)} x launches of {os.path.relpath(st_path, ROOT)} (committed rocprofv3 passes; bytes not re-measured in this run) over this run's ms_per_step"}
        except Exception as e:
            roof["whole_step_traffic"] = {"note": f"failed: {e!r}"}
        if exch is not None:
            out["exchanges"] = exch
        if replicas_extra is not None:
            out["replicas"] = replicas_extra
        if world == 1 and not args.no_ajtai:
            out["ajtai"] = ajtai_extra()
            if "witness_commit" in mem_info:
                out["ajtai"].append(mem_info["witness_commit"])
        if world == 1 and args.chain > 0 and args.streams == 1 and args.ccs == "r1cs":
            out["ivc"] = [ivc_extra(wl.name, args.chain, local_rank, elapsed / args.steps * 1e3), ivc_extra(wl.name, args.chain, local_rank, elapsed / args.steps * 1e3, overlap=False)]
            if wl.name != "C2" and wl.ring == "goldilocks":
                out["ivc"].append(ivc_extra("C2", args.chain, local_rank))
        if world == 1 and not args.no_shard_model and args.streams == 1 and args.ccs == "r1cs" and wl.ring == "goldilocks" and wl.m >= (1 << 18):   # (a step worth sharding)
            try:   # the prediction the driver's SCALE run (strong scaling of ONE fold stream) can be checked against
                from latticefold_amd.shard_model import model_summary
                out["shard_model"] = model_summary(wl, (2, 4, 8), steps=4, warmup=2, device=local_rank, base_ms=elapsed / args.steps * 1e3)
            except Exception as e:
                out["shard_model"] = {"note": f"failed: {e!r}"}
        if world == 1 and not args.no_lfplus:
            try:
                out["lfplus"] = lfplus_extra()
            except Exception as e:   # a reported extra, outside the timed region
                out["lfplus"] = {"op": "PlusProver::prove", "ms": None, "note": f"failed: {e!r}"}
        if world > 1 and lfplus_sharded is not None:
            out["lfplus"] = lfplus_sharded
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
