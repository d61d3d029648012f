"""world_size-2 gloo test (CPU) of the N>1 bench path: rendezvous on 127.0.0.1, per-rank independent instances
(seed = rank), barrier + max-over-ranks clock, aggregate steps/s.  The GPU step is replaced by the CPU oracle's
linearization at toy size -- this exercises the distributed glue of bench.py, which is all that N>1 adds in round 1
("replicas")."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = textwrap.dedent('''
    import os, sys, time, json
    sys.path.insert(0, os.environ["LF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["LF_ROOT"], "tests"))
    import numpy as np, torch, torch.distributed as dist
    import lfo
    from latticefold_amd.workload import make_workload
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    wl = make_workload("T8", seed=rank)          # independent instance per rank
    inst = lfo.Instance(wl)
    f = inst.witness_from_w_ccs(wl.w_ccs)
    cm = lfo.ajtai_commit(wl.ajtai_matrix(), wl.kappa, wl.N, lfo.crt(f))
    cccs = np.concatenate([cm, wl.x_ccs])
    dist.barrier(); t0 = time.perf_counter()
    steps = 2
    digest = 0
    for _ in range(steps):
        lc, pr = inst.linearize(lfo.Transcript(), cccs, f)
        digest = int(pr[2, 0]) ^ int(pr[-1, 0])   # p_1(2) and the last u (p_1(0) = p_1(1) = 0 for a satisfied CCS)
    dist.barrier(); el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    dg = [None] * world
    dist.all_gather_object(dg, digest)
    if rank == 0:
        print(json.dumps({"value": world * steps / float(el.item()), "n_gpus": world, "digests": dg}))
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_gloo_replicas(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LF_ROOT=ROOT, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["digests"][0] != d["digests"][1]       # ranks really worked on different instances
