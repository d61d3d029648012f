"""The LatticeFold+ oracle's transcript-driven part (oracle/lfp_protocol.c; CPU only): pinned to what the reference holds -- the Frog Poseidon
table (rings/poseidon/frog.rs, via checksums), the challenge-set decoding KAT (rings/frog.rs:66-96) -- and checked for completeness /
soundness against its own restated verifiers on the shapes of the reference's tests (setchk.rs:355-495, rgchk.rs:340-433)."""
import json
import os

import numpy as np
import pytest

import lfp

KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))
P, D = lfp.P, lfp.D


def test_frog_poseidon_table_matches_reference_checksums():
    k = KATS["poseidon_frog_params"]
    assert k["same_literals_as_goldilocks"] and (k["full_rounds"], k["partial_rounds"], k["alpha"], k["rate"], k["capacity"]) == (8, 22, 7, 20, 4)
    ark, mds = lfp.poseidon_params()
    assert [int(x) for x in ark[:4]] == k["ark_first"] and [int(x) for x in mds[-4:]] == k["mds_last"]
    assert sum((i + 1) * int(v) for i, v in enumerate(ark)) % P == k["ark_checksum"]
    assert sum((i + 1) * int(v) for i, v in enumerate(mds)) % P == k["mds_checksum"]


def test_short_challenge_decoding_kat():
    k = KATS["frog_short_challenge"]
    assert [int(x) for x in lfp.short_challenge_from_bytes(k["bytes"])] == k["expected_coeffs"]


def test_transcript_is_a_duplex_sponge():
    """absorb / squeeze bookkeeping: clones agree, different absorbs diverge, squeeze_bytes uses 7 bytes per element"""
    t1 = lfp.Transcript()
    t1.absorb(np.arange(32, dtype=np.uint64).reshape(2, 16))
    t2 = t1.clone()
    assert t1.challenge() == t2.challenge()
    c = [t1.challenge() for _ in range(25)]
    assert len(set(c)) == 25 and all(x < P for x in c)
    b1, b2 = t1.squeeze_bytes(16), t2.squeeze_bytes(16)
    assert b1.shape == (16,) and not (b1 == b2).all()      # t2 is 25 challenges behind
    t3, t4 = lfp.Transcript(), lfp.Transcript()
    t3.absorb(np.ones((1, 16), dtype=np.uint64)); t4.absorb(np.full((1, 16), 2, dtype=np.uint64))
    assert t3.challenge() != t4.challenge()
    s = lfp.Transcript().short_challenge()
    assert all(int(x) < 128 or int(x) >= P - 128 for x in s)


def test_psi_extracts_the_exponent():
    """ct(psi * exp(a)) = a for -d/2 < a < d/2 (LatticeFold+ Lemma 2.2; the property Dcom::verify relies on, rgchk.rs:199-205)"""
    psi = lfp.psi()
    for a in range(-7, 8):
        m = lfp.exp_dense(np.array([a], dtype=np.int8))[0]
        assert int(lfp.ring_mul(psi, m)[0]) == a % P


def _ident(n, first=None):
    rowptr = np.arange(n + 1, dtype=np.uint32)
    col = np.arange(n, dtype=np.uint32)
    val = np.zeros((n, D), dtype=np.uint64)
    val[:, 0] = 1
    if first is not None:
        val[0, 0] = first
    return rowptr, col, val


@pytest.mark.parametrize("shape", ["one", "batched", "mix", "mix_with_M"])
def test_set_check_completeness_and_soundness(shape):
    """setchk.rs:355-495: identity matrices (unit monomials X^0 on the diagonal, absent entries elsewhere), vectors of ones / X^2"""
    n, nvars = 4, 2
    eye = np.zeros((n, n, D), dtype=np.uint64)
    eye[np.arange(n), np.arange(n), 0] = 1
    nmat = 1 if shape == "one" else 2
    msets = np.stack([eye] * nmat)
    vsets = None
    if shape.startswith("mix"):
        v0 = np.zeros((n, D), dtype=np.uint64); v0[:, 0] = 1
        v1 = np.zeros((n, D), dtype=np.uint64); v1[:, 2] = 1
        vsets = np.stack([v0, v1])
    mats = [_ident(n, first=2)] if shape == "mix_with_M" else []
    out = lfp.set_check(lfp.Transcript(), nvars, msets, vsets, mats)
    assert (int(out["msgs"][0, 0, 0]) + int(out["msgs"][0, 1, 0])) % P == 0 and out["msgs"][0].any()   # claimed sum 0; the round polynomial itself is not zero
    rc, r = lfp.set_check_verify(lfp.Transcript(), nvars, out, nM=len(mats))
    assert rc == 0 and (r == out["r"]).all()
    # e[0][set][j] = sum_i eq(r, i) M[i][j]: for the identity, eq(r, j) in the constant coefficient
    assert all(int(out["e"][0, 0, j, 1:].sum()) == 0 for j in range(n))
    # a non-monomial entry (1 + X) is rejected (test_set_check_bad / _batched_bad / _mix_bad)
    bad = msets.copy(); bad[nmat - 1, 0, 0, 1] = 1
    outb = lfp.set_check(lfp.Transcript(), nvars, bad, vsets, mats)
    assert lfp.set_check_verify(lfp.Transcript(), nvars, outb, nM=len(mats))[0] == -3
    if vsets is not None and shape == "mix":
        badv = vsets.copy(); badv[0, 0, 1] = 1
        outv = lfp.set_check(lfp.Transcript(), nvars, msets, badv, mats)
        assert lfp.set_check_verify(lfp.Transcript(), nvars, outv)[0] == -3
    # tampered evaluation / message
    t = {k: v.copy() for k, v in out.items()}; t["e"][0, 0, 1, 0] ^= 1
    assert lfp.set_check_verify(lfp.Transcript(), nvars, t, nM=len(mats))[0] != 0
    t = {k: v.copy() for k, v in out.items()}; t["msgs"][0, 0, 0] = 1
    assert lfp.set_check_verify(lfp.Transcript(), nvars, t, nM=len(mats))[0] != 0


def _instance(n, kappa, k, seed):
    A = lfp.splitmix(seed, 0, kappa * n * D).reshape(kappa, n, D)
    v = (lfp.splitmix(seed + 1, 0, n * D) % np.uint64(63)).astype(np.int64) - 31          # |coefficients| < 8^2 / 2: two base-8 digits
    f = np.where(v < 0, np.uint64(P) - (-v).astype(np.uint64), v.astype(np.uint64)).reshape(n, D)
    l = 22                                                                                 # ceil(log_8 p)
    rg = lfp.rg_from_f(f, A, D // 2, k, l)
    tau_i = np.array([int(t) if int(t) <= P // 2 else int(t) - P for t in rg["tau"]], dtype=np.int64)
    return {"Mf": lfp.exp_dense(rg["Df"]), "tau": rg["tau"], "mtau": lfp.exp_dense(tau_i.astype(np.int8)), "f": f}


@pytest.mark.parametrize("L,nM", [(1, 0), (1, 1), (2, 1)])
def test_range_check_completeness_and_soundness(L, nM):
    """rgchk.rs:340-433 (test_range_check, test_range_check_mm) on random small-norm witnesses; two instances as Mlin::mlin folds them"""
    n, nvars, kappa, k = 1 << 14, 14, 1, 2                     # (tau = kappa k d l d = 11264 digits must fit n: rgchk.rs "small n unsupported")
    insts = [_instance(n, kappa, k, 10 + 5 * i) for i in range(L)]
    mats = [_ident(n, first=2)] * nM
    d = lfp.range_check(lfp.Transcript(), nvars, insts, k, mats)
    rc, r = lfp.range_check_verify(lfp.Transcript(), nvars, d, k)
    assert rc == 0 and (r == d["r"]).all()
    assert (d["v"] == d["c"][:, 0]).all()                    # "v is equal to c[0]" (rgchk.rs:123)
    # (c[0] is only absorbed by Dcom::verify -- rgchk.rs:241-251 compares v for ni = 0 and c[ni] for the M_i rows -- so tamper c where it is checked)
    for key, idx in (("a", (0, 0)), ("bb", (0, 0, 3)), ("v", (0, 5)), ("e", (nM, 1, 3, 1))) + ((("c", (0, nM, 2)),) if nM else ()):
        t = {kk: vv.copy() for kk, vv in d.items()}
        t[key][idx] = (int(t[key][idx]) + 1) % P
        assert lfp.range_check_verify(lfp.Transcript(), nvars, t, k)[0] != 0, key


def _cm_instance(n, kappa, k, ell, A, seed):
    v = (lfp.splitmix(seed + 1, 0, n * D) % np.uint64(63)).astype(np.int64) - 31
    f = np.where(v < 0, np.uint64(P) - (-v).astype(np.uint64), v.astype(np.uint64)).reshape(n, D)
    rg = lfp.rg_from_f(f, A, D // 2, k, ell)
    tau_i = np.array([int(t) if int(t) <= P // 2 else int(t) - P for t in rg["tau"]], dtype=np.int8)
    return {"Mf": lfp.exp_dense(rg["Df"]), "tau": rg["tau"], "mtau": lfp.exp_dense(tau_i), "f": f, "comMf": rg["comMf"],
            "fcoms": np.stack([rg["cm_f"], rg["C_Mf"], rg["cm_mtau"]])}


@pytest.mark.parametrize("L,nM,kappa,nvars", [(1, 0, 1, 14), (1, 1, 2, 15), (2, 1, 1, 14)])
def test_cm_prove_completeness_and_soundness(L, nM, kappa, nvars):
    """cm.rs:606-666 (test_com: n = 2^15, kappa 2, one matrix) and the two-instance shape Mlin::mlin feeds it: the restated CmProof::verify accepts,
    recomputes the same ComX, and rejects tampered proofs; the double-commitment identity sum_x tau(x) t(x) = tensor(c) . comh holds"""
    n, k, ell = 1 << nvars, 2, 22
    A = lfp.splitmix(3, 0, kappa * n * D).reshape(kappa, n, D)
    insts = [_cm_instance(n, kappa, k, ell, A, 30 + 7 * i) for i in range(L)]
    mats = [_ident(n, first=2)] * nM
    pr = lfp.cm_prove(lfp.Transcript(), nvars, insts, k, ell, kappa, mats)
    rc, x = lfp.cm_verify(lfp.Transcript(), pr, [i["fcoms"] for i in insts])
    assert rc == 0
    for key in ("cm_g", "ro", "vo"):
        assert (x[key] == pr[key]).all(), key
    # g = s0 tau + s1 m_tau + s2 f + h commits to cm_g (the homomorphism the fold relies on): A g = cm_g
    for l in range(L):
        assert (lfp.commit(A, pr["g"][l]) == pr["cm_g"][l]).all()
    for key, idx in (("comh", (0, 0, 1)), ("pa", (2, 1, 3)), ("pb", (0, 0, 0)), ("ea", (0, 2, 5)), ("eb", (L - 1, 3, 0)), ("a", (0, 0)), ("e", (0, 1, 2, 3))):
        t = dict(pr)
        t[key] = pr[key].copy()
        t[key][idx] = (int(t[key][idx]) + 1) % P
        assert lfp.cm_verify(lfp.Transcript(), t, [i["fcoms"] for i in insts])[0] != 0, key


def _r1cs_identity(n, k, B):
    m = n // k
    return lfp.r1cs_decomposed_square((lfp.identity_csr(m),) * 3, n, B, k)


def test_r1cs_linearize_completeness_and_soundness():
    """r1cs.rs:186-233 (test_linearization: identity constraint system, z = 1, gadget-decomposed), restated on the Frog ring with a 0/1 witness (z z = z)"""
    n, nvars, k, b = 1 << 7, 7, 4, 2
    r1cs = _r1cs_identity(n, k, b)
    z = np.zeros((n // k, D), dtype=np.uint64)
    z[::3, 0] = 1
    f = lfp.gadget_decompose(z, b, k)
    pr = lfp.r1cs_linearize(lfp.Transcript(), nvars, f, r1cs)
    rc, ro = lfp.r1cs_verify(lfp.Transcript(), pr)
    assert rc == 0 and (ro == pr["r"]).all()
    # an unsatisfied system (z = 2: 2 * 2 != 2) is rejected: the claimed sum is no longer zero
    z[:, 0] = 2
    bad = lfp.r1cs_linearize(lfp.Transcript(), nvars, lfp.gadget_decompose(z, 8, k), _r1cs_identity(n, k, 8))
    assert lfp.r1cs_verify(lfp.Transcript(), bad)[0] == -1
    for key, idx, want in (("msgs", (0, 0, 0), -1), ("msgs", (3, 2, 5), -1), ("evals", (1, 0), -2), ("evals", (3, 2), -2)):
        t = dict(pr)
        t[key] = pr[key].copy()
        t[key][idx] = (int(t[key][idx]) + 1) % P
        assert lfp.r1cs_verify(lfp.Transcript(), t)[0] == want, key


def test_plus_prove_and_verify_two_rounds():
    """plus.rs:148-272 (test_prove / test_prove_multi) at a size the CPU restatement folds in seconds: kappa 1, k 4, n 2^15; two fresh instances in the
    first round, one more folded into the accumulator in the second"""
    from math import ceil, log, sqrt
    n, k, kappa, L = 1 << 15, 4, 1, 3
    a, c = 16 * 128 * L, 8 + 16 * k + 1                     # utils::estimate_bound(sop, L, d, k) (utils.rs:102-112)
    B = ceil((a + sqrt(a * a + 4 * a * c)) / 2) // 2
    l = ceil(log(P) / log(8))
    A = lfp.splitmix(21, 0, kappa * n * D).reshape(kappa, n, D)
    r1cs = _r1cs_identity(n, k, B)
    rng = np.random.default_rng(5)

    def comp():
        z = np.zeros((n // k, D), dtype=np.uint64)
        z[:, 0] = rng.integers(0, 2, size=n // k)
        return lfp.gadget_decompose(z, B, k), r1cs

    prover, ts_v = lfp.PlusOracle(A, list(r1cs), kappa, 8, k, l, B, lfp.Transcript()), lfp.Transcript()
    for rnd, ncomp in enumerate((2, 1)):
        proof = prover.prove([comp() for _ in range(ncomp)])
        assert proof["cmproof"]["b"].shape[0] == (2 if rnd else 0) + ncomp
        assert (lfp.commit(A, proof["g"]) == proof["linb2x"]["cm_g"]).all()
        if rnd == 0:
            t = dict(proof, dproof=dict(proof["dproof"], C0=(proof["dproof"]["C0"] + np.uint64(1)) % np.uint64(P)))
            assert lfp.plus_verify(ts_v.clone(), t, B) == ("dproof", -1)
        assert lfp.plus_verify(ts_v, proof, B) == 0
