"""LatticeFold+ `PlusProver::prove` column-sharded over 2 / 4 ranks (BASELINE configs[4]): every rank must return, field by field, the proof
of the committed ORACLE-ONLY fixtures (tests/golden/lfplus_digests.json) -- at the reference's end-to-end bench shapes and at 2^20 rows -- and, for shapes
without a fixture, the proof of the unsharded run.  Ranks share cuda:0 here and exchange through gloo (host transport, lfplus_set_sharding); on a multi-GPU
node the same code runs one rank per GPU over the library's RCCL communicator (lfplus_dist_init): test_rccl_ranks_* below, skipped under `world` GPUs."""
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
    import os, sys, json
    sys.path.insert(0, os.environ["LF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["LF_ROOT"], "tests"))
    import numpy as np, torch, torch.distributed as dist
    from latticefold_amd import plus, dist as lfd
    from test_gpu_lfplus_scale import digests
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("LF_BACKEND", "gloo")
    dev = rank if os.environ.get("LF_PER_RANK_DEVICE") else 0
    if backend == "nccl":
        torch.cuda.set_device(dev)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    out = {}
    for name in os.environ["LF_CASES"].split(","):
        if name.startswith("S"):          # "S14.2.2.1": nvars, L, k, kappa -- a shape without a fixture
            nv, L, k, kappa = (int(x) for x in name[1:].split("."))
            plus.PLUS_CONFIGS[name] = (nv, L, k, kappa)
        wl = plus.make_plus_workload(name)
        r1cs = wl.r1cs()
        def run(sharded, rounds):
            shard = None
            A = wl.ajtai_matrix()
            if sharded:
                if os.environ.get("LF_TRANSPORT", "host") == "rccl":
                    ids = [plus.dist_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    shard = (rank, world, ids[0])
                else:
                    shard = (rank, world, lfd.make_allgather(dist.new_group()))
                A = wl.ajtai_matrix(lfd.column_shard(wl.n, rank, world))
            prover = plus.PlusProver.init(A, list(r1cs), max(1, wl.L - 2), wl.params(), plus.PoseidonTranscript(), dev, shard)
            res = []
            try:
                for rnd in range(rounds):         # round 0 folds the L fresh instances, later rounds accumulate one more each (plus.rs:219-272)
                    zs = [wl.z(i) for i in range(wl.L)] if rnd == 0 else [wl.z(wl.L + rnd)]
                    comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, z, 1, wl.B, wl.k) for z in zs]
                    proof = prover.prove(comps)
                    d = digests(proof, prover.acc, prover.transcript.clone().get_challenge())
                    d["exchanges"] = prover.ctxs[0].dist_stats()[0]
                    res.append(d)
            finally:
                prover.close()
            return res
        rounds = int(os.environ.get("LF_ROUNDS", "1"))
        ref = run(False, rounds) if (os.environ.get("LF_WITH_REF") and (rank == 0 or os.environ.get("LF_PER_RANK_DEVICE"))) else None
        got = run(True, rounds)
        allg, allr = [None] * world, [None] * world
        dist.all_gather_object(allg, got)
        dist.all_gather_object(allr, ref)
        if rank == 0:
            out[name] = {"ranks": allg, "ref": allr}
    if rank == 0:
        print(json.dumps(out))
    if backend == "nccl":
        dist.barrier(device_ids=[dev])
    dist.destroy_process_group()
''')


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _gold():
    p = os.path.join(ROOT, "tests", "golden", "lfplus_digests.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def _launch(tmp_path, world, cases, env_extra, timeout):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LF_ROOT=ROOT, OMP_NUM_THREADS="2", LF_CASES=cases, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def _fields(d):
    return {k: v for k, v in d.items() if k not in ("oracle_seconds", "oracle_host", "workload", "first_words", "exchanges")}


@pytest.mark.parametrize("world,cases,rounds", [(2, "S14.2.2.1,S15.3.4.1", 3), (4, "S14.2.2.1", 2), (8, "S15.2.2.1", 1)])
def test_sharded_plus_prover_equals_unsharded(tmp_path, world, cases, rounds):
    """small shapes, several accumulating rounds (the F0 / F1 of one prove are the first two instances of the next): every rank = the unsharded prover"""
    d = _launch(tmp_path, world, cases, {"LF_WITH_REF": "1", "LF_ROUNDS": str(rounds)}, 900)
    for name in cases.split(","):
        ranks, ref = d[name]["ranks"], d[name]["ref"][0]
        assert len(ranks) == world and len(ref) == rounds
        for r, got in enumerate(ranks):
            for rnd in range(rounds):
                bad = [k for k in _fields(ref[rnd]) if got[rnd][k] != ref[rnd][k]]
                assert not bad, f"{name} x{world}, rank {r}, round {rnd}: fields differing from the unsharded prover: {bad}"
            assert got[0]["exchanges"] > 0 and ref[0]["exchanges"] == 0


@pytest.mark.parametrize("world,name", [(2, "P15"), (4, "P16"), (2, "P17"), (2, "P20"), (4, "P20")])
def test_sharded_plus_prover_matches_committed_oracle_digests(tmp_path, world, name):
    gold = _gold()
    if name not in gold:
        pytest.skip(f"no golden digest for {name}")
    d = _launch(tmp_path, world, name, {}, 1500)
    ranks = d[name]["ranks"]
    assert len(ranks) == world
    want = _fields(gold[name])
    for r, got in enumerate(ranks):
        bad = [k for k in want if got[0][k] != want[k]]
        assert not bad, f"{name} sharded x{world}, rank {r}: fields differing from the oracle fixture: {bad}"
        assert got[0]["exchanges"] > 0


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_sharded_plus_prover(tmp_path, world):
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (RCCL refuses two ranks on one device); {torch.cuda.device_count()} visible")
    gold = _gold()
    cases = "S14.2.2.1,P16,P20"
    d = _launch(tmp_path, world, cases, {"LF_BACKEND": "nccl", "LF_TRANSPORT": "rccl", "LF_PER_RANK_DEVICE": "1", "LF_WITH_REF": "1"}, 1500)
    for name in cases.split(","):
        ranks, refs = d[name]["ranks"], d[name]["ref"]
        assert len(ranks) == world
        for r, got in enumerate(ranks):
            bad = [k for k in _fields(refs[r][0]) if got[0][k] != refs[r][0][k]]
            assert not bad, f"{name} over {world} RCCL ranks, rank {r}: fields differing from the unsharded run: {bad}"
            if name in gold:
                bad = [k for k in _fields(gold[name]) if got[0][k] != gold[name][k]]
                assert not bad, f"{name} over {world} RCCL ranks, rank {r}: fields differing from the oracle fixture: {bad}"
            assert got[0]["exchanges"] > 0
