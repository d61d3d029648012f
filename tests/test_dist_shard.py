"""Intra-step sharding, both rings: a fold step sharded over 2 and 4 ranks must return, on EVERY rank, the bit-identical proof,
folded LCCCS and folded witness of the unsharded run (which test_gpu_parity pins to the oracle).  Ranks share cuda:0 here and
exchange through gloo; on a multi-GPU node the same code runs one rank per GPU over RCCL (bench.py --parallelism shard)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu

WORKER = textwrap.dedent('''
    import os, sys, json, hashlib
    sys.path.insert(0, os.environ["LF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["LF_ROOT"], "tests"))
    import numpy as np, torch.distributed as dist
    from latticefold_amd import api, dist as lfd
    from latticefold_amd.workload import make_workload
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    out = {}
    for name in os.environ["LF_CASES"].split(","):
        wl = make_workload(name.split("/")[0], ccs=name.split("/")[1]) if "/" in name else make_workload(name)   # "T12/multi": another constraint system
        def run(sharded):
            ctx = api.Context(0, ring=wl.ring)
            tr = lambda: api.PoseidonTranscript(ring=wl.ring)
            if sharded:
                lfd.init_sharding(ctx, rank, world, os.environ.get("LF_TRANSPORT", "host"))
            ctx.load_ccs(wl)
            scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
            wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
            cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
            lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
            lc2, w1, proof2 = api.NIFSProver.prove(ctx, lc, w0, cccs, wit, tr())   # chained step
            h = hashlib.sha256()
            for a in (cccs, acc, lc, proof, w0.f_coeff, lc2, proof2, w1.f_coeff):
                h.update(np.ascontiguousarray(a).tobytes())
            paths = ctx.fold_paths()
            ctx.close()
            return h.hexdigest() + ":%d" % paths
        ref = run(False) if rank == 0 else None
        got = run(True)
        allg = [None] * world
        dist.all_gather_object(allg, got)
        if rank == 0:
            out[name] = {"ref": ref, "ranks": allg}
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()
''')


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


# "big": the thresholds of the large-instance round modes (fused fix, look-up-table rounds 2-4) lowered so that a 2^12 / 2^14-row instance takes
# them -- in the sharded run on the ranks' pair slices, in the unsharded reference on whole tables
@pytest.mark.parametrize("world,cases,two_lanes", [(2, "T10,G5,T12", False), (4, "T12", False), (2, "T12", True), (2, "B8,B10,BDP", False), (4, "B14", False),
                                                   (2, "T12,T14", "big"), (4, "T14", "big"), (2, "T12", "plain"), (8, "T14", "big"),
                                                   (2, "T14", "gemm"), (8, "T14", "gemm"),
                                                   # rows that refer to a neighbour's (and, wrapping round, to rank 0's) columns, and four matrices: the z-space
                                                   # combinations of a rank cover the columns its rows of G read (shard_col_range), not just its own slice
                                                   (2, "T10/multi,T12/deg3", False), (4, "T12/multi", "big"),
                                                   # one host thread issues every exchange (the schedule of a transport with ONE channel, or LF_SHARD_TWO_LANES=0)
                                                   (2, "T12", "one"), (4, "T14", "one"),
                                                   # the hand-over sizes of round 5: off (only the 64-pairs-per-rank rule of the earlier rounds: the slices are gathered
                                                   # late) and large (gathered after the first fix)
                                                   (4, "T14", "deep"), (2, "T14", "early")])
def test_sharded_fold_step_equals_unsharded(tmp_path, world, cases, two_lanes):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LF_ROOT=ROOT, OMP_NUM_THREADS="2", LF_CASES=cases)
    if two_lanes == "big":
        env.update(LF_FOLD_LUT_MIN="128", LF_FOLD_FUSE_MIN="64", LF_FOLD_TAB_MIN="64")
    elif two_lanes == "gemm":   # rounds 1-3 of the folding sumcheck as int8 GEMMs on the ranks' pair slices (lf_sv_rounds.h), then the large-instance modes
        env.update(LF_FOLD_SV_MIN="64", LF_DOT_MIN="64", LF_FOLD_LUT_MIN="128", LF_FOLD_FUSE_MIN="64", LF_FOLD_TAB_MIN="64")   # (LF_DOT_MIN: the int8 inner products on the ranks' column slices, odd first columns included)
    elif two_lanes == "plain":
        env.update(LF_SHARD_PLAIN_ROUNDS="1", LF_FOLD_LUT_MIN="128", LF_FOLD_FUSE_MIN="64", LF_FOLD_TAB_MIN="64")
    elif two_lanes == "one":
        env["LF_SHARD_TWO_LANES"] = "0"
    elif two_lanes == "deep":
        env.update(LF_SHARD_LIN_MIN="0", LF_SHARD_FOLD_MIN="0", LF_FOLD_LUT_MIN="128", LF_FOLD_FUSE_MIN="64", LF_FOLD_TAB_MIN="64")
    elif two_lanes == "early":
        env.update(LF_SHARD_LIN_MIN="1048576", LF_SHARD_FOLD_MIN="1048576", LF_FOLD_LUT_MIN="128", LF_FOLD_FUSE_MIN="64", LF_FOLD_TAB_MIN="64")
    elif two_lanes:   # the threaded two-lane schedule with one exchange channel per lane, forced (it is the default when the transport has two channels, as here)
        env["LF_SHARD_TWO_LANES"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for name, r in d.items():
        assert len(r["ranks"]) == world
        assert all(x.split(":")[0] == r["ref"].split(":")[0] for x in r["ranks"]), (name, r)
        if two_lanes == "gemm":   # every rank and the unsharded reference ran rounds 1-3 as GEMMs (fold_paths mask behind the digest)
            assert r["ref"].endswith(":7") and all(x.endswith(":7") for x in r["ranks"]), r


def test_model_transport_runs_a_ranks_share_and_counts_its_exchanges():
    """lf_set_sharding_model (tools/shard_model.py): rank r of G with no peers.  The step must run to completion with the exchange count and the
    words of a real G-way run's rank (zeros stand in for the peers, so the result is not a proof and is not compared); world 1 is the plain context."""
    import numpy as np
    from latticefold_amd import api
    from latticefold_amd.workload import make_workload
    wl = make_workload("T14")

    def run(rank, world):
        ctx = api.Context(0)
        try:
            ctx.set_sharding_model(rank, world)
            ctx.load_ccs(wl)
            scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
            wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
            cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
            tr = api.PoseidonTranscript()
            acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr)
            ctx.dist_stats(reset=True); ctx.dist_stats_words(reset=True)
            lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr)
            return ctx.dist_stats()[0], ctx.dist_stats_words(), proof
        finally:
            ctx.close()

    n1, w1, p1 = run(0, 1)
    assert n1 == 0 and w1 == 0
    ctx = api.Context(0)
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        tr = api.PoseidonTranscript()
        acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr)
        _, _, p0 = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr)
    finally:
        ctx.close()
    assert (p0 == p1).all()                       # world 1: no model, the ordinary step
    counts = [run(r, 4) for r in (0, 3)]
    assert counts[0][0] == counts[1][0] > 8 and counts[0][1] == counts[1][1] > 0   # every rank issues the same exchanges with the same sizes
    ctx = api.Context(0)
    try:
        with pytest.raises(api.LfError):
            ctx.set_sharding_model(4, 4)
    finally:
        ctx.close()
