"""The real bench.py on its N > 1 path, 2 ranks on ONE GPU (LF_FORCE_DEVICE pins both ranks to device 0, gloo replaces RCCL
because NCCL/RCCL refuses two ranks on one device): replicas (weak) and intra-step sharding (strong); checks the JSON
contract fields the driver reads."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("mode,workload", [("replicas", "T12"), ("shard", "T14"), ("replicas", "B10"), ("shard", "B14")])
def test_bench_two_ranks_one_gpu(mode, workload):
    env = dict(os.environ, LF_FORCE_DEVICE="0", LF_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", workload, "--parallelism", mode]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["scaling"] == ("weak" if mode == "replicas" else "strong")
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    # value is the whole-job aggregate: replicas count both ranks' steps
    per_rank = d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)
    assert abs(d["value"] - (2 if mode == "replicas" else 1) * per_rank) / d["value"] < 1e-6


def test_bench_streams_mode_single_rank():
    """opt-in throughput mode: 2 independent fold streams on one GPU from one process"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--workload", "T12", "--streams", "2",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and "2 independent streams" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-6
