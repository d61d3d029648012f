"""Column-sharded Ajtai commitment with all-gather + modular sum.
CPU: world_size-2 gloo, partial commitments from the oracle -> exercises the exchange + lf_modsum (host code of the product).
GPU: two ranks sharing cuda:0 (gloo for the tiny exchange), partials from the HIP kernel on column slices."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, os.environ["LF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["LF_ROOT"], "tests"))
    import numpy as np, torch.distributed as dist
    import lfo
    from latticefold_amd import api, dist as lfd
    from latticefold_amd.workload import make_workload, splitmix_fq
    use_gpu = os.environ["LF_USE_GPU"] == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    wl = make_workload("T10")
    A = wl.ajtai_matrix()                                  # every rank can regenerate any slice of the stream
    f = splitmix_fq(77, 0, 3 * wl.N * 24).reshape(3, wl.N, 24)
    lo, hi = lfd.column_shard(wl.N, rank, world)
    if use_gpu:
        ctx = api.Context(0)
        sch = api.AjtaiCommitmentScheme(ctx, matrix=np.ascontiguousarray(A[:, lo:hi]))
        part = sch.commit_ntt(np.ascontiguousarray(f[:, lo:hi]))
    else:
        part = np.stack([lfo.ajtai_commit(np.ascontiguousarray(A[:, lo:hi]), wl.kappa, hi - lo, np.ascontiguousarray(f[b, lo:hi])) for b in range(3)])
    full = lfd.allgather_modsum(part)
    want = np.stack([lfo.ajtai_commit(A, wl.kappa, wl.N, f[b]) for b in range(3)])
    ok = bool((full == want).all()) and not bool((part == want).all())
    res = [None] * world
    dist.all_gather_object(res, ok)
    if rank == 0:
        print(json.dumps({"ok": all(res), "world": world}))
    dist.destroy_process_group()
''')


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(tmp_path, use_gpu):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LF_ROOT=ROOT, OMP_NUM_THREADS="2", LF_USE_GPU="1" if use_gpu else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["ok"] and d["world"] == 2


def test_sharded_commit_gloo_cpu(tmp_path):
    _run(tmp_path, use_gpu=False)


@pytest.mark.gpu
def test_sharded_commit_two_ranks_one_gpu(tmp_path):
    _run(tmp_path, use_gpu=True)


def test_modsum_rejects_non_canonical():
    from latticefold_amd import api
    P = api.P
    parts = np.array([[1, P - 1], [P - 1, 5]], dtype=np.uint64)
    out = np.zeros(2, dtype=np.uint64)
    assert api._lib().lf_modsum(parts.ctypes.data_as(api.u64p), 2, 2, out.ctypes.data_as(api.u64p)) == 0
    assert out.tolist() == [0, 4]
    bad = np.array([[P, 0]], dtype=np.uint64)
    assert api._lib().lf_modsum(bad.ctypes.data_as(api.u64p), 1, 2, out.ctypes.data_as(api.u64p)) == -1
    # the BabyBear modulus through the ring-aware entry point
    PB = 15 * 2**27 + 1
    parts = np.array([[1, PB - 1], [PB - 1, 5]], dtype=np.uint64)
    assert api._lib().lf_modsum_ring(parts.ctypes.data_as(api.u64p), 2, 2, out.ctypes.data_as(api.u64p), 1) == 0
    assert out.tolist() == [0, 4]
    bad = np.array([[PB, 0]], dtype=np.uint64)
    assert api._lib().lf_modsum_ring(bad.ctypes.data_as(api.u64p), 1, 2, out.ctypes.data_as(api.u64p), 1) == -1
    assert api._lib().lf_modsum_ring(parts.ctypes.data_as(api.u64p), 2, 2, out.ctypes.data_as(api.u64p), 7) == -1
