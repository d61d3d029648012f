"""Host-only part of the LatticeFold+ slice of liblfhip.so (no GPU needed): the Frog PoseidonTranscript and the set-check / range-check
VERIFIERS, against the reference-held KATs and against the oracle (oracle/lfp_protocol.c): the product's transcript produces the oracle's
challenges, and the product's verifiers accept the oracle's proofs and reject tampered ones at the stage the oracle reports."""
import json
import os

import numpy as np
import pytest

import lfp
from latticefold_amd import plus

KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))
P, D = plus.P, plus.D


def test_exported_symbols_include_the_protocol_slice():
    lib = plus._lib()
    for s in ("lfplus_transcript_new", "lfplus_set_check", "lfplus_set_check_verify", "lfplus_range_check", "lfplus_range_check_verify", "lfplus_short_challenge"):
        assert s in plus.exported_symbols() and hasattr(lib, s)


def test_product_poseidon_table_matches_reference_checksums():
    k = KATS["poseidon_frog_params"]
    ark, mds = plus.poseidon_params()
    assert sum((i + 1) * int(v) for i, v in enumerate(ark)) % P == k["ark_checksum"]
    assert sum((i + 1) * int(v) for i, v in enumerate(mds)) % P == k["mds_checksum"]
    assert [int(x) for x in ark[:4]] == k["ark_first"]


def test_optimised_permutation_equals_the_definition_and_the_oracle():
    """the transcript runs the permutation in Montgomery form with the partial rounds factored sparse: same output as the textbook form and as the oracle's"""
    rng = np.random.default_rng(11)
    L_ = lfp._plib()
    states = [np.zeros(24, dtype=np.uint64), np.full(24, P - 1, dtype=np.uint64)] + [rng.integers(0, P, size=24, dtype=np.uint64) for _ in range(50)]
    for st in states:
        want = st.copy()
        L_.lfp_poseidon_permute(lfp._p(want))
        assert (plus.poseidon_permute(st) == want).all() and (plus.poseidon_permute(st, plain=True) == want).all()
        assert (plus.poseidon_permute(st, plain=2) == want).all()


def test_ifma_lanes_of_the_frog_permutation_equal_the_scalar_forms():
    """lfp_poseidon_simd.cc (AVX-512 IFMA, Montgomery words with R = 2^104): the form the transcript runs on hosts that have the instructions.  Sparse states,
    words next to p, long chains (a wrong carry shows up after a few permutations) -- against the textbook definition and the scalar sparse form; and a whole
    transcript script gives the same challenges with LFPLUS_POSEIDON_SCALAR=1 in a fresh process."""
    rng = np.random.default_rng(5)
    states = []
    for k in range(300):
        st = rng.integers(0, P, size=24, dtype=np.uint64)
        if k % 5 == 0:
            st[rng.integers(0, 24, size=12)] = 0
        if k % 7 == 0:
            st[rng.integers(0, 24, size=8)] = np.uint64(P - 1 - (k % 3))
        if k % 11 == 0:
            st[:] = np.uint64(k & 1)
        states.append(st)
    for st in states:
        a = plus.poseidon_permute(st)
        assert (a == plus.poseidon_permute(st, plain=2)).all()
    for st in states[:40]:
        assert (plus.poseidon_permute(st) == plus.poseidon_permute(st, plain=True)).all()
    st = states[1].copy()
    ref = st.copy()
    for _ in range(2000):
        st = plus.poseidon_permute(st)
        ref = plus.poseidon_permute(ref, plain=2)
    assert (st == ref).all()
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import numpy as np; from latticefold_amd import plus\n"
            "t = plus.PoseidonTranscript(); rng = np.random.default_rng(3)\n"
            "out = []\n"
            "for i in range(40):\n"
            "    t.absorb(rng.integers(0, plus.P, size=(1 + i %% 5, 16), dtype=np.uint64)); out.append(int(t.get_challenge()))\n"
            "print(int(plus.poseidon_simd()), out)\n") % os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    runs = []
    for env in ({}, {"LFPLUS_POSEIDON_SCALAR": "1"}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, **env}, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(r.stdout.strip().split(" ", 1))
    assert runs[1][0] == "0" and runs[0][1] == runs[1][1]
    if not plus.poseidon_simd():
        pytest.skip("this host has no AVX-512 IFMA: the scalar form is what runs (compared with itself)")


def test_transcript_equals_oracle_on_random_scripts():
    rng = np.random.default_rng(3)
    tp, to = plus.PoseidonTranscript(), lfp.Transcript()
    for step in range(60):
        op = rng.integers(0, 4)
        if op == 0:
            x = lfp.splitmix(100 + step, 0, int(rng.integers(1, 4)) * D).reshape(-1, D)
            tp.absorb(x); to.absorb(x)
        elif op == 1:
            assert tp.get_challenge() == to.challenge()
        elif op == 2:
            n = int(rng.integers(1, 40))
            assert (tp.squeeze_bytes(n) == to.squeeze_bytes(n)).all()
        else:
            assert (tp.short_challenge() == to.short_challenge()).all()
    assert tp.clone().get_challenge() == to.clone().challenge()


def _ident(n, first=None):
    rowptr, col = np.arange(n + 1, dtype=np.uint32), np.arange(n, dtype=np.uint32)
    val = np.zeros((n, D), dtype=np.uint64)
    val[:, 0] = 1
    if first is not None:
        val[0, 0] = first
    return rowptr, col, val


def test_set_check_verifier_accepts_oracle_proofs_and_rejects_tampering():
    n, nvars = 8, 3
    rng = np.random.default_rng(1)
    dig = rng.integers(-7, 8, size=(2, n, 4)).astype(np.int8)
    vdig = rng.integers(-7, 8, size=(1, n)).astype(np.int8)
    mats = [_ident(n, first=3)]
    out = lfp.set_check(lfp.Transcript(), nvars, lfp.exp_dense(dig), lfp.exp_dense(vdig), mats)
    ok, stage, r = plus.set_check_verify(plus.PoseidonTranscript(), nvars, out, nM=1)
    assert ok and stage == 0 and (r == out["r"]).all()
    for key, idx in (("e", (0, 1, 2, 5)), ("b", (0, 3)), ("msgs", (1, 2, 0)), ("msgs", (0, 0, 4))):
        t = {k: v.copy() for k, v in out.items()}
        t[key][idx] = (int(t[key][idx]) + 1) % P
        ok_p, st_p, _ = plus.set_check_verify(plus.PoseidonTranscript(), nvars, t, nM=1)
        rc_o, _ = lfp.set_check_verify(lfp.Transcript(), nvars, t, nM=1)
        assert not ok_p and st_p == -rc_o, (key, st_p, rc_o)


def test_range_check_verifier_accepts_oracle_proofs_and_rejects_tampering():
    n, nvars, kappa, k = 1 << 14, 14, 1, 2
    A = lfp.splitmix(5, 0, kappa * n * D).reshape(kappa, n, D)
    v = (lfp.splitmix(6, 0, n * D) % np.uint64(63)).astype(np.int64) - 31
    f = np.where(v < 0, np.uint64(P) - (-v).astype(np.uint64), v.astype(np.uint64)).reshape(n, D)
    rg = lfp.rg_from_f(f, A, D // 2, k, 22)
    tau_i = np.array([int(t) if int(t) <= P // 2 else int(t) - P for t in rg["tau"]], dtype=np.int8)
    inst = {"Mf": lfp.exp_dense(rg["Df"]), "tau": rg["tau"], "mtau": lfp.exp_dense(tau_i), "f": f}
    d = lfp.range_check(lfp.Transcript(), nvars, [inst], k, [_ident(n, first=2)])
    d.update(k=k, nvars=nvars)
    ok, stage, r = plus.range_check_verify(plus.PoseidonTranscript(), d)
    assert ok and stage == 0 and (r == d["r"]).all()
    for key, idx, want in (("a", (0, 1), 4), ("bb", (0, 0, 2), 4), ("v", (0, 7), 5), ("c", (0, 1, 9), 5), ("e", (0, 0, 3, 1), 3)):
        t = dict(d)
        t[key] = d[key].copy()
        t[key][idx] = (int(t[key][idx]) + 1) % P
        ok_p, st_p, _ = plus.range_check_verify(plus.PoseidonTranscript(), t)
        assert not ok_p and st_p == want, (key, st_p)
        assert lfp.range_check_verify(lfp.Transcript(), nvars, t, k)[0] == -want


@pytest.mark.parametrize("nvars,kappa", [(14, 1), (16, 4)])     # kappa 4: tensor(c) has two variables (their order matters: utils.rs:118-131)
def test_cm_verifier_accepts_oracle_proofs_and_rejects_tampering(nvars, kappa):
    """CmProof::verify (cm.rs:349-543) of the product (host only) on proofs the oracle's Cm::prove wrote: same verdict, same stage, same folded instance"""
    k, ell, L = 2, 22, 2
    n = 1 << nvars
    A = lfp.splitmix(3, 0, kappa * n * D).reshape(kappa, n, D)
    insts = []
    for i in range(L):
        v = (lfp.splitmix(31 + 7 * i, 0, n * D) % np.uint64(63)).astype(np.int64) - 31
        f = np.where(v < 0, np.uint64(P) - (-v).astype(np.uint64), v.astype(np.uint64)).reshape(n, D)
        rg = lfp.rg_from_f(f, A, D // 2, k, ell)
        tau_i = np.array([int(t) if int(t) <= P // 2 else int(t) - P for t in rg["tau"]], dtype=np.int8)
        insts.append({"Mf": lfp.exp_dense(rg["Df"]), "tau": rg["tau"], "mtau": lfp.exp_dense(tau_i), "f": f, "comMf": rg["comMf"],
                      "fcoms": np.stack([rg["cm_f"], rg["C_Mf"], rg["cm_mtau"]])})
    fcoms = [i["fcoms"] for i in insts]
    pr = lfp.cm_prove(lfp.Transcript(), nvars, insts, k, ell, kappa, [_ident(n, first=2)])
    tp = plus.PoseidonTranscript()
    ok, stage, x = plus.cm_verify(tp, pr, fcoms)
    assert ok and stage == 0
    for key in ("cm_g", "ro", "vo"):
        assert (x[key] == pr[key]).all(), key
    to = lfp.Transcript()
    assert lfp.cm_verify(to, pr, fcoms)[0] == 0 and tp.get_challenge() == to.challenge()
    for key, idx in (("comh", (0, 0, 1)), ("pa", (2, 1, 3)), ("pb", (0, 0, 0)), ("ea", (0, 2, 5)), ("eb", (1, 3, 0)), ("a", (0, 0)), ("e", (0, 1, 2, 3)),
                     ("pa", (nvars - 1, 2, 15))):
        t = dict(pr)
        t[key] = pr[key].copy()
        t[key][idx] = (int(t[key][idx]) + 1) % P
        ok_p, st_p, _ = plus.cm_verify(plus.PoseidonTranscript(), t, fcoms)
        rc_o = lfp.cm_verify(lfp.Transcript(), t, fcoms)[0]
        assert not ok_p and st_p == -rc_o, (key, st_p, rc_o)
    # kappa' k d l d > n: "t0 too large!" (cm.rs:601) -> stage 7 on both sides
    t = dict(pr, ell=64)
    assert plus.cm_verify(plus.PoseidonTranscript(), t, fcoms)[1] == 7 and lfp.cm_verify(lfp.Transcript(), t, fcoms)[0] == -7


def test_plus_verifier_accepts_oracle_proofs_and_rejects_tampering():
    """PlusVerifier::verify (plus.rs:134-146) of the product, host only, over two rounds of the oracle's PlusProver: ComR1CSProof::verify, CmProof::verify and
    DecompProof::verify agree with the oracle's verdicts and stages"""
    from math import ceil, log, sqrt
    n, k, kappa, L = 1 << 15, 4, 1, 3
    a, c = 16 * 128 * L, 8 + 16 * k + 1
    B = ceil((a + sqrt(a * a + 4 * a * c)) / 2) // 2
    l = ceil(log(P) / log(8))
    A = lfp.splitmix(22, 0, kappa * n * D).reshape(kappa, n, D)
    r1cs = lfp.r1cs_decomposed_square((lfp.identity_csr(n // k),) * 3, n, B, k)
    # the product's host-side R1CS helpers build the same matrices and the same gadget decomposition
    mine = plus.r1cs_decomposed_square((plus.identity_csr(n // k),) * 3, n, B, k)
    for m0, m1 in zip(r1cs, mine):
        assert all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(m0, m1))
    rng = np.random.default_rng(6)
    zs = []
    for _ in range(3):
        z = np.zeros((n // k, D), dtype=np.uint64)
        z[:, 0] = rng.integers(0, 2, size=n // k)
        zs.append(z)
    zr = rng.integers(0, P, size=(64, D), dtype=np.uint64)
    assert (plus.gadget_decompose(zr, B, k) == lfp.gadget_decompose(zr, B, k)).all()
    assert (plus.gadget_decompose(zr, 8, 22) == lfp.gadget_decompose(zr, 8, 22)).all()
    prover = lfp.PlusOracle(A, list(r1cs), kappa, 8, k, l, B, lfp.Transcript())
    params = plus.PlusParameters(plus.LinParameters(kappa, plus.DecompParameters(8, k, l)), B)
    ver, ts_o = plus.PlusVerifier.init(A, list(r1cs), params, plus.PoseidonTranscript()), lfp.Transcript()
    for rnd, comps in enumerate(([zs[0], zs[1]], [zs[2]])):
        proof = prover.prove([(lfp.gadget_decompose(z, B, k), r1cs) for z in comps])
        if rnd == 0:
            for where, key, idx, want in (("lproof", "msgs", (2, 1, 0), ("lproof[1]", 1)), ("lproof", "evals", (2, 3), ("lproof[1]", 2)),
                                          ("cmproof", "pb", (4, 1, 2), ("cmproof", 6)), ("dproof", "v1", (1, 0, 3), ("dproof", 2)),
                                          ("dproof", "C1", (0, 5), ("dproof", 1))):
                t = dict(proof)
                if where == "lproof":
                    lp = dict(proof["lproof"][1])
                    lp[key] = lp[key].copy()
                    lp[key][idx] = (int(lp[key][idx]) + 1) % P
                    t["lproof"] = [proof["lproof"][0], lp]
                else:
                    sub = dict(proof[where])
                    sub[key] = sub[key].copy()
                    sub[key][idx] = (int(sub[key][idx]) + 1) % P
                    t[where] = sub
                v2 = plus.PlusVerifier.init(A, list(r1cs), params, plus.PoseidonTranscript())
                assert not v2.verify(t) and v2.stage == want, (where, key, v2.stage)
                got = lfp.plus_verify(lfp.Transcript(), t, B)
                assert got == (want[0], -want[1]), got
        assert ver.verify(proof) and lfp.plus_verify(ts_o, proof, B) == 0
    assert ver.transcript.get_challenge() == ts_o.challenge()


def test_verifiers_refuse_malformed_proofs_instead_of_reading_past_them():
    """The host verifiers are the one component that sees untrusted input: the C entry points index raw pointers with the shape parameters they are given, so
    the wrappers must size every array from the VERIFIER's parameters and refuse (LFPLUS_E_ARG / stage "malformed") what does not fit -- never read it."""
    z16 = lambda *s: np.zeros(s + (D,), dtype=np.uint64)
    # ComR1CSProof with nvars = 32 but one round of messages
    with pytest.raises(plus.LfPlusError) as e:
        plus.r1cs_verify(plus.PoseidonTranscript(), {"nvars": 32, "msgs": z16(1, 4), "evals": z16(4)})
    assert e.value.code == plus.E_ARG
    with pytest.raises(plus.LfPlusError):
        plus.r1cs_verify(plus.PoseidonTranscript(), {"nvars": 5, "msgs": z16(5, 4), "evals": z16(4)}, nvars=9)       # the verifier's nvars wins
    # Out with a one-element e against nM = 100000
    with pytest.raises(plus.LfPlusError) as e:
        plus.set_check_verify(plus.PoseidonTranscript(), 4, {"e": z16(1, 1, 1), "b": z16(0), "msgs": z16(4, 4)}, nM=100000)
    assert e.value.code == plus.E_ARG
    # Dcom / CmProof: every field is checked against (nvars, L, k, nM, kappa)
    nvars, L, k, nM, kappa = 6, 2, 2, 1, 1
    per = 4 + 4 * nM
    good = {"nvars": nvars, "k": k, "ell": 2, "kappa": kappa, "msgs": z16(nvars, 4), "e": z16(1 + nM, L * k, D), "b": z16(L), "v": z16(L),
            "a": np.zeros((L, 1 + nM), dtype=np.uint64), "bb": z16(L, 1 + nM), "c": z16(L, 1 + nM), "comh": z16(L, kappa), "pa": z16(nvars, 3),
            "pb": z16(nvars, 3), "ea": z16(L, per), "eb": z16(L, per)}
    fcoms = [z16(3, kappa)] * L
    assert plus.cm_verify(plus.PoseidonTranscript(), good, fcoms, nvars=nvars, L=L, k=k, ell=2, kappa=kappa, nM=nM)[0] is False      # well-formed, wrong: a verdict
    for key in ("msgs", "e", "b", "v", "a", "bb", "c", "comh", "pa", "pb", "ea", "eb"):
        bad = dict(good)
        bad[key] = good[key][:-1] if good[key].shape[0] > 1 else good[key][..., :-1]
        with pytest.raises(plus.LfPlusError) as e:
            plus.cm_verify(plus.PoseidonTranscript(), bad, fcoms, nvars=nvars, L=L, k=k, ell=2, kappa=kappa, nM=nM)
        assert e.value.code == plus.E_ARG, key
    with pytest.raises(plus.LfPlusError):
        plus.cm_verify(plus.PoseidonTranscript(), good, fcoms, nvars=nvars, L=L, k=k, ell=2, kappa=kappa, nM=3)          # more matrices on the verifier's side
    with pytest.raises(plus.LfPlusError):
        plus.range_check_verify(plus.PoseidonTranscript(), dict(good, nvars=20))                                          # metadata larger than the arrays
    with pytest.raises(plus.LfPlusError):
        plus.decomp_verify({"C0": z16(1), "C1": z16(1), "v0": z16(1, 2), "v1": z16(1, 2)}, z16(2), z16(1, 2), 7)
    # PlusVerifier takes n, kappa from ITS matrix, k / ell from ITS parameters and nM from ITS matrix list
    n = 1 << nvars
    params = plus.PlusParameters(plus.LinParameters(kappa, plus.DecompParameters(8, k, 2)), 7)
    ver = plus.PlusVerifier.init(np.zeros((kappa, n, D), dtype=np.uint64), [None] * nM, params, plus.PoseidonTranscript())
    lp = {"nvars": 32, "msgs": z16(1, 4), "evals": z16(4)}
    assert not ver.verify({"lproof": [lp], "cmproof": dict(good, fcoms=np.stack(fcoms)), "dproof": {}, "linb2x": {}}) and ver.stage[0] == "malformed"
    cm_bad = dict(good, fcoms=np.stack(fcoms), nvars=3, k=9, kappa=7, ell=1)                                               # lying metadata is ignored: shapes rule
    ver = plus.PlusVerifier.init(np.zeros((kappa, n, D), dtype=np.uint64), [None] * nM, params, plus.PoseidonTranscript())
    assert not ver.verify({"lproof": [], "cmproof": cm_bad, "dproof": {"C0": z16(kappa), "C1": z16(kappa), "v0": z16(1 + nM, 2), "v1": z16(1 + nM, 2)},
                           "linb2x": {"cm_g": z16(kappa), "vo": z16(1 + nM, 2)}}) and ver.stage[0] == "cmproof"
    ver = plus.PlusVerifier.init(np.zeros((kappa, n, D), dtype=np.uint64), [None] * (nM + 2), params, plus.PoseidonTranscript())
    assert not ver.verify({"lproof": [], "cmproof": dict(good, fcoms=np.stack(fcoms)), "dproof": {}, "linb2x": {}}) and ver.stage[0] == "malformed"
    with pytest.raises(plus.LfPlusError):
        plus.PlusVerifier.init(np.zeros((kappa + 1, n, D), dtype=np.uint64), [], params, plus.PoseidonTranscript())
