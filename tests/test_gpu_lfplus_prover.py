"""LatticeFold+ end to end on the GPU (crates/latticefold-plus/src/{r1cs,lin,mlin,plus}.rs): ComR1CS::linearize, Mlin::mlin and PlusProver::prove against
the oracle's restatement, field by field; the product's host-only PlusVerifier and the oracle's verifier accept the proofs."""
from math import ceil, log, sqrt

import numpy as np
import pytest

import lfp
from latticefold_amd import plus

pytestmark = pytest.mark.gpu
D, P = 16, plus.P


def _bound(L, k):
    a, c = 16 * 128 * L, 8 + 16 * k + 1                  # utils::estimate_bound (utils.rs:102-112)
    return ceil((a + sqrt(a * a + 4 * a * c)) / 2)


def _same(got, want, keys, where):
    for key in keys:
        assert (np.asarray(got[key]) == np.asarray(want[key])).all(), (where, key)


CM_KEYS = ("msgs", "r", "e", "b", "v", "a", "bb", "c", "comh", "pa", "ea", "pb", "eb", "ro", "cm_g", "vo", "fcoms")


@pytest.mark.parametrize("nvars,k,b,unfused", [(7, 4, 2, False), (12, 2, 8, False), (12, 2, 8, True), (13, 2, 8, True)])
def test_r1cs_linearize_matches_oracle(nvars, k, b, unfused, monkeypatch):
    """r1cs.rs:186-233 (test_linearization) and a larger shape; a ring-coefficient matrix makes the products genuinely polynomial.
    unfused: LFPLUS_CM_UNFUSED=1 (read per call) -- fix_variables as its own pass instead of deferred into the next round's kernel (k_r1cs_round_fused)"""
    if unfused:
        monkeypatch.setenv("LFPLUS_CM_UNFUSED", "1")
    else:
        monkeypatch.delenv("LFPLUS_CM_UNFUSED", raising=False)
    n = 1 << nvars
    r1cs = list(plus.r1cs_decomposed_square((plus.identity_csr(n // k),) * 3, n, b, k))
    rng = np.random.default_rng(3)
    r1cs[0][2][1, :] = rng.integers(0, P, size=D, dtype=np.uint64)        # (the relation need not hold for parity)
    z = rng.integers(0, P, size=(n // k, D), dtype=np.uint64)
    A = lfp.splitmix(2, 0, n * D).reshape(1, n, D)
    ctx = plus.PlusContext(0)
    try:
        cr = plus.ComR1CS.new(ctx, r1cs, z, 1, b, k, A)
        assert (cr.f == lfp.gadget_decompose(z, b, k)).all() and (cr.cm_f == lfp.commit(A, cr.f)).all()
        to, tp = lfp.Transcript(), plus.PoseidonTranscript()
        want = lfp.r1cs_linearize(to, nvars, cr.f, r1cs)
        linb, got = cr.linearize(ctx, tp)
        _same(got, want, ("msgs", "r", "evals"), "lproof")
        assert tp.get_challenge() == to.challenge()
        ok, st, _ = plus.r1cs_verify(plus.PoseidonTranscript(), got)
        assert (ok, st) == (False, 1) and lfp.r1cs_verify(lfp.Transcript(), got)[0] == -1      # the random system is not satisfied
    finally:
        ctx.close()


_PLUS_ORACLE = {}


@pytest.mark.parametrize("kappa,k,rounds,device_acc", [(2, 2, (2,), False), (1, 4, (2, 1, 1), False), (1, 4, (2, 1, 1), True)])
def test_plus_prover_matches_oracle(kappa, k, rounds, device_acc):
    """plus.rs:148-272: test_prove (n = 2^15, kappa 2, k 2, two fresh instances, one round) and the accumulating shape of test_prove_multi (k 4; kappa 1
    keeps tau inside n = 2^15) over three rounds.  device_acc: the witnesses are preloaded (PlusProver.preload) and the accumulator (F0, F1) stays on the
    device between the proves (lfplus_decompose_resident; read back with lfplus_get_witness) -- the same proofs and the same accumulator"""
    n, L = 1 << 15, 3
    B = _bound(L, k) + 1 if k == 2 else _bound(L, k) // 2
    l = ceil(log(P) / log(8))
    A = lfp.splitmix(23, 0, kappa * n * D).reshape(kappa, n, D)
    r1cs = plus.r1cs_decomposed_square((plus.identity_csr(n // k),) * 3, n, B, k)
    rng = np.random.default_rng(8)
    params = plus.PlusParameters(plus.LinParameters(kappa, plus.DecompParameters(8, k, l)), B)
    case = (kappa, k, rounds)
    if case not in _PLUS_ORACLE:      # the oracle's proofs, accumulators and closing challenge: one run per shape (device_acc changes nothing the oracle sees)
        oracle = lfp.PlusOracle(A, list(r1cs), kappa, 8, k, l, B, lfp.Transcript())
        rng_o, runs = np.random.default_rng(8), []
        for ncomp in rounds:
            zs_o = []
            for _ in range(ncomp):
                z = np.zeros((n // k, D), dtype=np.uint64)
                z[:, 0] = rng_o.integers(0, 2, size=n // k)
                zs_o.append(z)
            want = oracle.prove([(lfp.gadget_decompose(z, B, k), r1cs) for z in zs_o])
            runs.append((want, [np.array(a, copy=True) for a in oracle.acc]))
        _PLUS_ORACLE[case] = (runs, oracle.tr.challenge())
    runs, closing = _PLUS_ORACLE[case]
    prover = plus.PlusProver.init(A, list(r1cs), 1, params, plus.PoseidonTranscript())
    prover.device_acc = device_acc
    ver, ts_o = plus.PlusVerifier.init(A, list(r1cs), params, plus.PoseidonTranscript()), lfp.Transcript()
    try:
        for ncomp, (want, want_acc) in zip(rounds, runs):
            zs = []
            for _ in range(ncomp):
                z = np.zeros((n // k, D), dtype=np.uint64)
                z[:, 0] = rng.integers(0, 2, size=n // k)
                zs.append(z)
            comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, z, 1, B, k) for z in zs]
            if device_acc:
                prover.preload(comps)
            got = prover.prove(comps)
            for i in range(ncomp):
                _same(got["lproof"][i], want["lproof"][i], ("msgs", "r", "evals"), f"lproof[{i}]")
            _same(got["cmproof"], want["cmproof"], CM_KEYS, "cmproof")
            _same(got["linb2x"], want["linb2x"], ("cm_g", "ro", "vo"), "linb2x")
            _same(got["dproof"], want["dproof"], ("C0", "C1", "v0", "v1"), "dproof")
            acc = prover.accumulator()
            assert not device_acc or prover.acc == ["ctx0", "ctx1"]
            for i in range(2):
                assert (acc[i] == want_acc[i]).all()
            assert ver.verify(got), ver.stage
            assert lfp.plus_verify(ts_o, got, B) == 0
        assert prover.transcript.get_challenge() == closing
    finally:
        prover.close()
        if device_acc:             # the contexts' blocks went to the process-wide scratch cache: release them once here (later provers allocate afresh)
            plus.scratch_trim(0)


def test_scratch_cache_is_bounded_and_released(monkeypatch):
    """lfp_ctx.h LfpDevCache: a destroyed context leaves its scratch in the process-wide cache (so that a prover per proof does not pay hipMalloc again), the
    cache is bounded by min(32 GB, a quarter of the device's memory) per device, lfplus_scratch_bytes reports it and lfplus_scratch_trim gives it back to the
    driver -- the main path's allocators call the same trim before they report out-of-memory (lf_common.h lf_dev_malloc)"""
    from latticefold_amd import api
    plus.scratch_trim(0)
    assert plus.scratch_bytes(0) == 0
    n = 1 << 14
    rng = np.random.default_rng(5)
    main = api.Context(0)                    # (the main path's context: lf_device_memory = hipMemGetInfo)
    try:
        ctx = plus.PlusContext(0)
        try:
            ctx.set_witness(rng.integers(0, 31, size=(n, D), dtype=np.uint64))
        finally:
            ctx.close()
        held = plus.scratch_bytes(0)
        free0, total = main.device_memory()
        assert n * D * 8 <= held <= min(32 << 30, total // 4)
        plus.scratch_trim(0)
        assert plus.scratch_bytes(0) == 0
        assert main.device_memory()[0] >= free0 + held - (64 << 20)      # the blocks went back to the driver (other allocations may move a little)
    finally:
        main.close()


def test_set_witness_checks_the_words_on_the_device_and_prover_uploads_in_the_background(monkeypatch):
    """lfplus_set_witness checks canonicity behind the upload (k_check_canonical): a word >= p anywhere fails the call with LFPLUS_E_ARG and leaves no resident
    witness.  PlusProver uploads host witnesses on a worker thread while it linearizes the ones that have arrived: same proof as with LFPLUS_SERIAL_UPLOADS=1
    and as with preloaded witnesses; an upload that fails surfaces as the prove's error."""
    n = 1 << 12
    rng = np.random.default_rng(9)
    ctx = plus.PlusContext(0)
    try:
        good = rng.integers(0, 31, size=(n, D), dtype=np.uint64)
        ctx.set_witness(good)
        ctx.n = n                                       # (no matrix in this context: get_witness sizes its buffer from it)
        assert (ctx.get_witness() == good).all()
        for pos in ((0, 0), (n - 1, D - 1), (n // 2 + 1, 3)):
            bad = good.copy()
            bad[pos] = np.uint64(plus.P + (pos[1] % 2))
            with pytest.raises(plus.LfPlusError) as e:
                ctx.set_witness(bad)
            assert e.value.code == plus.E_ARG
            with pytest.raises(plus.LfPlusError):
                ctx.get_witness()                       # nothing resident after a refused upload
        ctx.set_witness(good)
        assert (ctx.get_witness() == good).all()
    finally:
        ctx.close()
    wl = plus.make_plus_workload("P15")
    A, r1cs = wl.ajtai_matrix(), wl.r1cs()

    def run(mode):
        if mode == "serial":
            monkeypatch.setenv("LFPLUS_SERIAL_UPLOADS", "1")
        else:
            monkeypatch.delenv("LFPLUS_SERIAL_UPLOADS", raising=False)
        prover = plus.PlusProver.init(A, list(r1cs), max(1, wl.L - 2), wl.params(), plus.PoseidonTranscript(), 0)
        try:
            comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, wl.z(i), 1, wl.B, wl.k) for i in range(wl.L)]
            if mode == "preload":
                prover.preload(comps)
            p1 = prover.prove(comps)
            p2 = prover.prove(comps[:1])              # second fold: the host accumulator (F0, F1) goes up on the worker thread as well
            return p1, p2, prover.accumulator(), prover.transcript.get_challenge()
        finally:
            prover.close()

    def flat(x):
        if isinstance(x, dict):
            return [v for k in sorted(x) for v in flat(x[k])]
        if isinstance(x, (list, tuple)):
            return [v for y in x for v in flat(y)]
        return [np.asarray(x)]
    ref = run("thread")
    for mode in ("serial", "preload"):
        got = run(mode)
        assert got[3] == ref[3]
        for a, b in zip(flat(ref[:3]), flat(got[:3])):
            assert a.shape == b.shape and (a == b).all(), mode
    prover = plus.PlusProver.init(A, list(r1cs), max(1, wl.L - 2), wl.params(), plus.PoseidonTranscript(), 0)
    try:
        comps = [plus.ComR1CS.new(prover.ctxs[0], r1cs, wl.z(i), 1, wl.B, wl.k) for i in range(wl.L)]
        comps[1].f = comps[1].f.copy()
        comps[1].f[5, 5] = np.uint64(plus.P)
        with pytest.raises(plus.LfPlusError):
            prover.prove(comps)
        with pytest.raises(plus.LfPlusError):
            prover.prove(comps)                        # a failed prover refuses to continue
    finally:
        prover.close()
