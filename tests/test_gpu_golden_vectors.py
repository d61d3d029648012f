"""The product against COMMITTED oracle fixtures only (tests/golden/abi_vectors.json from tests/tools/make_abi_vectors.py,
tests/golden/kats.json): no oracle library is loaded here.  Per C-ABI entry point at toy sizes, plus complete fold steps (T8 word for
word, C1 = BASELINE configs[0] and B6 by section digests)."""
import hashlib
import json
import os

import numpy as np
import pytest

from latticefold_amd import api
from latticefold_amd.workload import make_workload, splitmix_fq

pytestmark = pytest.mark.gpu
V = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abi_vectors.json")))


def arr(x, *shape):
    return np.array(x, dtype=np.uint64).reshape(*shape)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


@pytest.mark.parametrize("ring", ["goldilocks", "babybear"])
def test_entry_points_against_fixtures(ring):
    d = V[ring]
    RE, TAU = d["RE"], d["TAU"]
    rnd = lambda seed, n: splitmix_fq(seed, 0, n * RE, ring).reshape(n, RE)
    ctx = api.Context(0, ring=ring)
    try:
        x = rnd(d["crt"]["seed"], d["crt"]["count"])
        assert (ctx.crt(x) == arr(d["crt"]["out"], -1, RE)).all()
        assert (ctx.icrt(x) == arr(d["icrt"]["out"], -1, RE)).all()
        k = d["decompose"]
        e = arr(k["input"], -1, RE)
        assert (ctx.decompose(e, k["base"], k["digits"], 0) == arr(k["layout0"], -1, RE)).all()
        assert (ctx.decompose(e, k["base"], k["digits"], 1) == arr(k["layout1"], -1, RE)).all()
        k = d["recompose"]
        sm = (rnd(k["seed"], k["count_out"] * k["digits"]) % np.uint64(k["mod"])).astype(np.uint64)
        assert (ctx.recompose(sm, k["base"], k["digits"]) == arr(k["out"], -1, RE)).all()
        k = d["ajtai_commit"]
        A = splitmix_fq(k["seed_A"], 0, k["kappa"] * k["n"] * RE, ring).reshape(k["kappa"], k["n"], RE)
        f = rnd(k["seed_f"], k["n"])
        assert (api.AjtaiCommitmentScheme(ctx, matrix=A).commit_ntt(f) == arr(k["out"], -1, RE)).all()
        k = d["build_eq"]
        pt = splitmix_fq(k["seed"], 0, k["nv"] * TAU, ring).reshape(k["nv"], TAU)
        assert (ctx.build_eq(pt) == arr(k["out_slot0"], -1, TAU)).all()
        k = d["mle_eval"]
        tb = rnd(k["seed_table"], k["len"])
        got = ctx.evaluate_mles(tb[None, :, :], pt)
        assert (got.reshape(-1) == arr(k["out"], -1)).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("key", ["fold_T8", "fold_C1", "fold_B6"])
def test_fold_step_against_fixtures(key):
    d = V[key]
    wl = make_workload(d["workload"])
    ctx = api.Context(0, ring=wl.ring)
    try:
        ctx.load_ccs(wl)
        scheme = api.AjtaiCommitmentScheme(ctx, kappa=wl.kappa, n=wl.N, seed=wl.ajtai_seed())
        wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
        tr = lambda: api.PoseidonTranscript(ring=wl.ring)
        cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
        acc, lin = api.LFLinearizationProver.prove(ctx, cccs, wit, tr())
        lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr())
        lcs, dec = api.LFDecompositionProver.prove(ctx, acc, wit, tr())
        got = {"f_coeff": sha(wit.f_coeff), "cccs": sha(cccs), "acc": sha(acc), "lin_proof": sha(lin), "lcccs_out": sha(lc), "f0_ntt": sha(w0.f),
               "proof": sha(proof), "dec_proof_of_acc": sha(dec), "dec_lcccs_of_acc": sha(lcs)}
        bad = [k for k in d["sha"] if got[k] != d["sha"][k]]
        assert not bad, bad
        if "proof" in d:
            assert (proof.reshape(-1) == arr(d["proof"], -1)).all() and (lc.reshape(-1) == arr(d["lcccs_out"], -1)).all()
            # NIFSVerifier on the fixture proof (host verifier of the product, no GPU object involved)
            ok, lc_v, stage = api.NIFSVerifier.verify(wl, arr(d["acc"], -1, wl.RE), arr(d["cccs"], -1, wl.RE), arr(d["proof"], -1, wl.RE), tr())
            assert ok and (lc_v.reshape(-1) == arr(d["lcccs_out"], -1)).all(), stage
    finally:
        ctx.close()
