"""CPU side of the committed fixtures: the oracle (rebuilt from oracle/*.c) still reproduces tests/golden/abi_vectors.json, i.e. the
fixtures the GPU tests compare against are the oracle's, and the product's host verifier accepts the fixture proof."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
V = json.load(open(os.path.join(HERE, "golden", "abi_vectors.json")))


def test_oracle_reproduces_the_committed_vectors(tmp_path):
    """regenerate into a scratch file with the committed script and compare (element-wise vectors and T8 in full, C1 / B6 digests)"""
    script = os.path.join(HERE, "tools", "make_abi_vectors.py")
    src = open(script).read().replace('OUT = os.path.join(ROOT, "tests", "golden", "abi_vectors.json")', f'OUT = {str(tmp_path / "v.json")!r}')
    p = tmp_path / "gen.py"
    p.write_text(src.replace("HERE = os.path.dirname(os.path.abspath(__file__))", f"HERE = {os.path.join(HERE, 'tools')!r}"))
    out = subprocess.run([sys.executable, str(p)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.load(open(tmp_path / "v.json")) == V


def test_host_verifier_accepts_the_fixture_proof():
    sys.path.insert(0, os.path.join(HERE, ".."))
    from latticefold_amd import api
    from latticefold_amd.workload import make_workload
    d = V["fold_T8"]
    wl = make_workload("T8")
    a = lambda k: np.array(d[k], dtype=np.uint64).reshape(-1, wl.RE)
    ok, lc, stage = api.NIFSVerifier.verify(wl, a("acc"), a("cccs"), a("proof"), api.PoseidonTranscript())
    assert ok and (lc == a("lcccs_out")).all(), stage
    bad = a("proof").copy()
    bad[5, 1] ^= np.uint64(1)
    ok, _, stage = api.NIFSVerifier.verify(wl, a("acc"), a("cccs"), bad, api.PoseidonTranscript())
    assert not ok and stage == 1
