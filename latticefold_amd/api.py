"""Host-side mirror of the reference's interface for the prover hot path, over the C ABI of liblfhip.so.

Names follow crates/latticefold: `AjtaiCommitmentScheme` (commitment/commitment_scheme.rs:17-114),
`Witness` (arith.rs:213-362), `PoseidonTranscript` (transcript/poseidon.rs), `DecompositionParams`
(decomposition_parameters.rs:11-20), `NIFSProver.prove` (nifs.rs:48-103), `CCS` (arith.rs:50-74).
All bulk data are numpy uint64 arrays of canonical residues, shape (..., d) per ring element: d = 24 for
GoldilocksRingNTT (default), d = 72 for BabyBearRingNTT (`Context(device, ring="babybear")`, `PoseidonTranscript(ring=..)`).

There is NO CPU fallback: if the HIP library is missing or no GPU is present every call raises.
"""
import ctypes as C
import os

import numpy as np

RE = 24
P = 2**64 - 2**32 + 1
RING_IDS = {"goldilocks": 0, "babybear": 1}   # LF_RING_* of include/lfhip.h
RING_WORDS = {"goldilocks": 24, "babybear": 72}
RING_TAU = {"goldilocks": 3, "babybear": 9}
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblfhip.so")

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, u64p, u64p, C.c_size_t)   # lf_exchange_fn


class LfError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = _lib().lf_strerror(code).decode() if _LIB is not None else str(code)
        super().__init__(f"liblfhip {where}: {msg} ({code})")


class CommitmentError(LfError):
    """CommitmentError::WrongWitnessLength (commitment.rs:14-27)"""


class Params(C.Structure):
    """lf_params == DecompositionParams {B, L, B_SMALL=b, K} + CCS shape + kappa."""
    _fields_ = [("s", C.c_uint32), ("wit_len", C.c_uint32), ("l", C.c_uint32), ("L", C.c_uint32),
                ("K", C.c_uint32), ("b", C.c_uint32), ("B", C.c_uint64), ("kappa", C.c_uint32),
                ("t", C.c_uint32), ("q", C.c_uint32), ("d", C.c_uint32)]


_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f"{_SO} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is required; there is no CPU fallback)")
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.lf_strerror.restype = C.c_char_p
        L.lf_strerror.argtypes = [C.c_int]
        L.lf_ctx_create.argtypes = [C.POINTER(vp), C.c_int]
        L.lf_ctx_create_ring.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
        L.lf_ctx_ring.argtypes = [vp]
        L.lf_ring_modulus.restype = C.c_uint64
        L.lf_ring_modulus.argtypes = [C.c_int]
        for f in ("lf_lcccs_len_ring", "lf_cccs_len_ring", "lf_proof_len_ring"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.POINTER(Params), C.c_int]
        L.lf_transcript_new_ring.restype = vp
        L.lf_transcript_new_ring.argtypes = [C.c_int]
        L.lf_poseidon_params_ring.argtypes = [u64p, u64p, C.c_int]
        L.lf_poseidon_params_ring.restype = None
        L.lf_proof_wire_size.argtypes = [C.POINTER(Params), C.c_int]
        L.lf_proof_wire_size.restype = C.c_size_t
        L.lf_proof_serialize.argtypes = [C.POINTER(Params), C.c_int, u64p, C.c_void_p, C.c_size_t]
        L.lf_proof_deserialize.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_size_t, u64p]
        L.lf_poseidon_permute_ring.argtypes = [u64p, C.c_int, C.c_int]
        L.lf_poseidon_permute_ring.restype = None
        L.lf_ctx_destroy.argtypes = [vp]
        L.lf_ctx_destroy.restype = None
        L.lf_set_ring_tables.argtypes = [vp, C.c_uint64, u64p]
        L.lf_set_ext_basis.argtypes = [vp, u64p]
        L.lf_set_digit_mode.argtypes = [vp, C.c_int]
        L.lf_get_ring_tables.argtypes = [vp, u64p, u64p]
        L.lf_device_synchronize.argtypes = [vp]
        L.lf_mem_info.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.lf_selftest_field.argtypes = [vp, C.c_uint64, C.c_uint32, u64p]
        L.lf_ntt_fwd.argtypes = [vp, u64p, u64p, C.c_size_t]
        L.lf_ntt_inv.argtypes = [vp, u64p, u64p, C.c_size_t]
        L.lf_decompose.argtypes = [vp, u64p, C.c_size_t, C.c_uint64, C.c_uint, C.c_int, u64p]
        L.lf_recompose.argtypes = [vp, u64p, C.c_size_t, C.c_uint64, C.c_uint, u64p]
        L.lf_linf_check.argtypes = [vp, u64p, C.c_size_t, C.c_uint64, C.c_int, C.POINTER(C.c_int), u64p]
        L.lf_ajtai_load.argtypes = [vp, u64p, C.c_size_t, C.c_size_t]
        L.lf_ajtai_generate.argtypes = [vp, C.c_uint64, C.c_size_t, C.c_size_t]
        L.lf_device_memory.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.lf_ajtai_commit.argtypes = [vp, u64p, C.c_size_t, C.c_size_t, u64p]
        L.lf_modsum.argtypes = [u64p, C.c_size_t, C.c_size_t, u64p]
        L.lf_modsum_ring.argtypes = [u64p, C.c_size_t, C.c_size_t, u64p, C.c_int]
        L.lf_set_sharding.argtypes = [vp, C.c_int, C.c_int, EXCHANGE_FN, vp]
        L.lf_set_sharding_lanes.argtypes = [vp, C.c_int, C.c_int, EXCHANGE_FN, vp, EXCHANGE_FN, vp]
        L.lf_dist_unique_id.argtypes = [C.c_void_p]
        L.lf_dist_init.argtypes = [vp, C.c_int, C.c_int, C.c_void_p]
        L.lf_dist_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        L.lf_build_eq.argtypes = [vp, u64p, C.c_uint, u64p]
        L.lf_mle_eval_batch.argtypes = [vp, u64p, C.c_size_t, C.c_size_t, u64p, C.c_uint, u64p]
        L.lf_ccs_load.argtypes = [vp, C.POINTER(Params), C.POINTER(u32p), C.POINTER(u32p), C.POINTER(u64p), u32p, u32p, u64p]
        L.lf_spmv.argtypes = [vp, C.c_uint, u64p, u64p]
        for f in ("lf_lcccs_len", "lf_cccs_len", "lf_proof_len"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.POINTER(Params)]
        L.lf_witness_from_w_ccs.argtypes = [vp, u64p, C.POINTER(vp)]
        L.lf_witness_from_f_coeff.argtypes = [vp, u64p, C.POINTER(vp)]
        L.lf_witness_from_f.argtypes = [vp, u64p, C.POINTER(vp)]
        L.lf_witness_get_f_coeff.argtypes = [vp, vp, u64p]
        L.lf_witness_get_f.argtypes = [vp, vp, u64p]
        L.lf_witness_get_w_ccs.argtypes = [vp, vp, u64p]
        L.lf_witness_commit.argtypes = [vp, vp, u64p]
        L.lf_witness_free.argtypes = [vp]
        L.lf_witness_free.restype = None
        L.lf_transcript_new.restype = vp
        L.lf_transcript_clone.restype = vp
        L.lf_transcript_clone.argtypes = [vp]
        L.lf_transcript_free.argtypes = [vp]
        L.lf_transcript_free.restype = None
        L.lf_transcript_absorb_fq.argtypes = [vp, u64p, C.c_size_t]
        L.lf_transcript_absorb_fq.restype = None
        L.lf_transcript_absorb_ring.argtypes = [vp, u64p, C.c_size_t]
        L.lf_transcript_absorb_ring.restype = None
        L.lf_transcript_get_challenge.argtypes = [vp, u64p]
        L.lf_transcript_get_challenge.restype = None
        L.lf_transcript_get_short_challenge.argtypes = [vp, u64p]
        L.lf_transcript_get_short_challenge.restype = None
        L.lf_poseidon_params.argtypes = [u64p, u64p]
        L.lf_poseidon_params.restype = None
        L.lf_poseidon_permute.argtypes = [u64p, C.c_int]
        L.lf_poseidon_permute.restype = None
        L.lf_sumcheck_lin_begin.argtypes = [vp, u64p, u64p]
        L.lf_sumcheck_lin_round.argtypes = [vp, u64p, u64p]
        L.lf_sumcheck_lin_end.argtypes = [vp]
        L.lf_linearize.argtypes = [vp, vp, u64p, vp, u64p, u64p]
        L.lf_fold_step.argtypes = [vp, vp, u64p, vp, u64p, vp, u64p, C.POINTER(vp), u64p]
        L.lf_device_sponge.argtypes = [vp, u32p, C.c_size_t, u64p, C.c_size_t, u64p, C.c_size_t, u64p]
        L.lf_sumcheck_fold_begin.argtypes = [vp, u64p, u64p]
        L.lf_sumcheck_fold_round.argtypes = [vp, u64p, u64p]
        L.lf_sumcheck_fold_end.argtypes = [vp]
        L.lf_lincomb.argtypes = [vp, u64p, u64p, C.c_size_t, C.c_size_t, u64p]
        L.lf_horner_combine.argtypes = [vp, u64p, C.c_size_t, C.c_size_t, C.c_size_t, u64p, u64p]
        L.lf_decomposition_prove.argtypes = [vp, vp, u64p, vp, u64p, u64p]
        L.lf_folding_prove.argtypes = [vp, vp, u64p, vp, vp, u64p, C.POINTER(vp), u64p]
        L.lf_last_phase_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.lf_phase_name.restype = C.c_char_p
        L.lf_phase_name.argtypes = [C.c_int]
        L.lf_verify_host.argtypes = [C.c_int, C.POINTER(Params), u32p, u32p, u64p, vp, u64p, u64p, u64p, u64p, C.POINTER(C.c_int)]
        L.lf_last_fold_paths.argtypes = [vp, C.POINTER(C.c_uint)]
        L.lf_last_lin_split_rounds.argtypes = [vp, C.POINTER(C.c_uint)]
        L.lf_last_fold_split_rounds.argtypes = [vp, C.POINTER(C.c_uint)]
        L.lf_last_timeline.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.c_int]
        L.lf_last_kernel_stats.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def dist_unique_ids():
    """two ncclUniqueIds (256 bytes) for Context.dist_init: one RCCL communicator per lane of the fold step"""
    out = b""
    for _ in range(2):
        buf = (C.c_uint8 * 128)()
        _chk(_lib().lf_dist_unique_id(buf), "lf_dist_unique_id")
        out += bytes(buf)
    return out


def abi_version():
    """LFHIP_ABI_VERSION the library was built with (include/lfhip.h)"""
    L = _lib()
    L.lf_abi_version.restype = C.c_int
    return int(L.lf_abi_version())


def exported_symbols():
    """Every symbol include/lfhip.h declares (used by the CPU-side ABI test)."""
    import re
    hdr = open(os.path.join(_HERE, "..", "include", "lfhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lf_[a-z0-9_]+)\s*\(", hdr)))


def _a64(x):
    a = np.ascontiguousarray(x, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


def _chk(rc, where):
    if rc != 0:
        raise LfError(rc, where)


class Context:
    """Owns the device memory (lf_ctx).  One per GPU / process."""

    def __init__(self, device=0, ring="goldilocks"):
        self.h = C.c_void_p()
        self.ring = ring
        self.ring_id = RING_IDS[ring]
        self.RE = RING_WORDS[ring]
        self.TAU = RING_TAU[ring]
        _chk(_lib().lf_ctx_create_ring(C.byref(self.h), device, self.ring_id), "lf_ctx_create_ring")
        self.params = None

    def close(self):
        if getattr(self, "h", None):
            _lib().lf_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ring tables -------------------------------------------------------------------
    def set_ring_tables(self, nonres, y):
        a, p = _a64(y)
        _chk(_lib().lf_set_ring_tables(self.h, nonres, p), "lf_set_ring_tables")

    def get_ring_tables(self):
        nr = C.c_uint64()
        y = np.zeros(8 * self.TAU, dtype=np.uint64)
        _chk(_lib().lf_get_ring_tables(self.h, C.cast(C.byref(nr), u64p), y.ctypes.data_as(u64p)), "lf_get_ring_tables")
        return nr.value, y

    def selftest_field(self, seed=1, n=1 << 20):
        m = C.c_uint64(1)
        _chk(_lib().lf_selftest_field(self.h, seed, n, C.cast(C.byref(m), u64p)), "lf_selftest_field")
        return m.value

    def set_digit_mode(self, mode):
        """lf_set_digit_mode: 0 = sign-magnitude truncation (default), 1 = floor rule (digits in [-base/2, base/2))"""
        _chk(_lib().lf_set_digit_mode(self.h, int(mode)), "lf_set_digit_mode")

    def set_ext_basis(self, T):
        """lf_set_ext_basis: T (tau x tau) maps the internal binomial-basis coordinates of F_{p^tau} to the caller's external ones"""
        a, p = _a64(np.ascontiguousarray(T, dtype=np.uint64).reshape(-1))
        assert a.size == self.TAU * self.TAU
        _chk(_lib().lf_set_ext_basis(self.h, p), "lf_set_ext_basis")

    def set_sharding(self, rank, world, allgather, allgather_lane1=None):
        """Intra-step sharding over a HOST transport (lf_set_sharding_lanes); call before creating the AjtaiCommitmentScheme.
        allgather(np.uint64[words]) -> np.uint64[world, words] in rank order (see latticefold_amd.dist.make_allgather).  The two lanes
        of a fold step exchange concurrently from two threads: pass a second callable with its own ordered channel (its own process
        group) as allgather_lane1."""
        def mk(fn):
            def _cb(user, send, recv, words):
                try:
                    mine = np.ctypeslib.as_array(send, shape=(words,)).copy()
                    out = np.ascontiguousarray(fn(mine), dtype=np.uint64).reshape(world * words)
                    C.memmove(recv, out.ctypes.data, world * words * 8)
                    return 0
                except Exception as e:  # never let an exception cross the C boundary
                    import sys
                    print("lf exchange callback failed:", repr(e), file=sys.stderr)
                    return -1
            return EXCHANGE_FN(_cb)
        self._exchange_cb = mk(allgather) if world > 1 else EXCHANGE_FN(0)
        self._exchange_cb1 = mk(allgather_lane1) if (world > 1 and allgather_lane1 is not None) else EXCHANGE_FN(0)
        _chk(_lib().lf_set_sharding_lanes(self.h, rank, world, self._exchange_cb, None, self._exchange_cb1, None), "lf_set_sharding_lanes")
        self.shard = (rank, world)

    def dist_init(self, rank, world, ids):
        """Intra-step sharding over RCCL (lf_dist_init): ids = bytes of TWO ncclUniqueIds (api.dist_unique_ids() on rank 0, broadcast by
        the launcher).  Exchanges then run on device buffers in the library's own streams (ncclAllGather + modular-sum kernel)."""
        ids = bytes(ids)
        assert len(ids) == 256
        buf = (C.c_uint8 * 256).from_buffer_copy(ids)
        _chk(_lib().lf_dist_init(self.h, rank, world, buf), "lf_dist_init")
        self.shard = (rank, world)

    def dist_two_lanes(self, set=-1):
        """lf_dist_two_lanes: 1 when sharded steps run the threaded two-lane schedule (the outcome of lf_dist_init's self-check); set = 0 / 1 overrides it --
        every rank must use the same value"""
        L = _lib()
        L.lf_dist_two_lanes.argtypes = [C.c_void_p, C.c_int]
        rc = L.lf_dist_two_lanes(self.h, int(set))
        if rc < 0:
            raise LfError(rc, "lf_dist_two_lanes")
        return rc

    def dist_stats(self, reset=False):
        """(number of exchanges, total us, max us) of the context's exchange log"""
        n, tot, mx = C.c_uint64(), C.c_double(), C.c_double()
        _chk(_lib().lf_dist_stats(self.h, C.byref(n), C.byref(tot), C.byref(mx), int(reset)), "lf_dist_stats")
        return n.value, tot.value, mx.value

    def set_sharding_model(self, rank, world):
        """lf_set_sharding_model: a TIMING model of rank `rank` of `world` with no peers (zeros stand in for their words).  What such a context returns is not
        a proof; tools/shard_model.py measures a rank's share of a sharded step with it on a one-GPU box."""
        L = _lib()
        L.lf_set_sharding_model.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _chk(L.lf_set_sharding_model(self.h, int(rank), int(world)), "lf_set_sharding_model")
        self.shard = (rank, world)

    def dist_stats_words(self, reset=False):
        """u64 words this rank contributed to its exchanges (lf_dist_stats_words)"""
        L = _lib()
        L.lf_dist_stats_words.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        w = C.c_uint64()
        _chk(L.lf_dist_stats_words(self.h, C.byref(w), int(reset)), "lf_dist_stats_words")
        return w.value

    def mem_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        _chk(_lib().lf_mem_info(self.h, C.byref(f), C.byref(t)), "lf_mem_info")
        return f.value, t.value

    def synchronize(self):
        _chk(_lib().lf_device_synchronize(self.h), "lf_device_synchronize")

    # ---- element-wise ops ----------------------------------------------------------------
    def crt(self, coeff):  # CRT::elementwise_crt
        a, p = _a64(coeff)
        o = np.empty_like(a)
        _chk(_lib().lf_ntt_fwd(self.h, p, o.ctypes.data_as(u64p), a.size // self.RE), "lf_ntt_fwd")
        return o

    def icrt(self, ntt):  # ICRT::elementwise_icrt
        a, p = _a64(ntt)
        o = np.empty_like(a)
        _chk(_lib().lf_ntt_inv(self.h, p, o.ctypes.data_as(u64p), a.size // self.RE), "lf_ntt_inv")
        return o

    def decompose(self, coeff, base, digits, layout):
        a, p = _a64(coeff)
        cnt = a.size // self.RE
        o = np.zeros((cnt * digits, self.RE), dtype=np.uint64)
        _chk(_lib().lf_decompose(self.h, p, cnt, base, digits, layout, o.ctypes.data_as(u64p)), "lf_decompose")
        return o

    def recompose(self, x, base, digits):
        a, p = _a64(x)
        cnt = a.size // self.RE // digits
        o = np.zeros((cnt, self.RE), dtype=np.uint64)
        _chk(_lib().lf_recompose(self.h, p, cnt, base, digits, o.ctypes.data_as(u64p)), "lf_recompose")
        return o

    def linf_check(self, f_ntt, bound, unsigned_variant=False):
        a, p = _a64(f_ntt)
        ok = C.c_int()
        mx = C.c_uint64()
        _chk(_lib().lf_linf_check(self.h, p, a.size // self.RE, bound, int(unsigned_variant), C.byref(ok), C.cast(C.byref(mx), u64p)), "lf_linf_check")
        return bool(ok.value), mx.value

    def build_eq(self, point):
        a, p = _a64(point)
        nv = a.size // self.TAU
        o = np.zeros(((1 << nv), self.TAU), dtype=np.uint64)
        _chk(_lib().lf_build_eq(self.h, p, nv, o.ctypes.data_as(u64p)), "lf_build_eq")
        return o

    def evaluate_mles(self, tables, point):  # utils/mle_helpers.rs:65-88
        a, p = _a64(tables)
        nt, ln = a.shape[0], a.shape[1]
        b, q = _a64(point)
        o = np.zeros((nt, self.RE), dtype=np.uint64)
        _chk(_lib().lf_mle_eval_batch(self.h, p, nt, ln, q, b.size // self.TAU, o.ctypes.data_as(u64p)), "lf_mle_eval_batch")
        return o

    # ---- CCS -------------------------------------------------------------------------------
    def load_ccs(self, wl):
        """wl: latticefold_amd.workload.Workload (params + CSR matrices)."""
        self.params = Params(wl.s, wl.wit_len, wl.l, wl.L, wl.K, wl.b, wl.B, wl.kappa, wl.t, wl.q, wl.d)
        t = wl.t
        rp = [np.ascontiguousarray(a, dtype=np.uint32) for a in wl.rowptr]
        ci = [np.ascontiguousarray(a, dtype=np.uint32) for a in wl.col]
        va = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1) for a in wl.val]
        so = np.ascontiguousarray(wl.S_off, dtype=np.uint32)
        si = np.ascontiguousarray(wl.S_idx, dtype=np.uint32)
        cc = np.ascontiguousarray(wl.c, dtype=np.uint64).reshape(-1)
        rpp = (u32p * t)(*[a.ctypes.data_as(u32p) for a in rp])
        cip = (u32p * t)(*[a.ctypes.data_as(u32p) for a in ci])
        vap = (u64p * t)(*[a.ctypes.data_as(u64p) for a in va])
        _chk(_lib().lf_ccs_load(self.h, C.byref(self.params), C.cast(rpp, C.POINTER(u32p)), C.cast(cip, C.POINTER(u32p)),
                                C.cast(vap, C.POINTER(u64p)), so.ctypes.data_as(u32p), si.ctypes.data_as(u32p),
                                cc.ctypes.data_as(u64p)), "lf_ccs_load")
        self.lcccs_len = _lib().lf_lcccs_len_ring(C.byref(self.params), self.ring_id)
        self.cccs_len = _lib().lf_cccs_len_ring(C.byref(self.params), self.ring_id)
        self.proof_len = _lib().lf_proof_len_ring(C.byref(self.params), self.ring_id)
        self.N = wl.N
        self.m = wl.m
        self.n = wl.n

    def mat_vec_mul(self, j, z):  # arith/utils.rs:52-65
        a, p = _a64(z)
        o = np.zeros((self.m, self.RE), dtype=np.uint64)
        _chk(_lib().lf_spmv(self.h, j, p, o.ctypes.data_as(u64p)), "lf_spmv")
        return o

    def phase_ms(self):
        out = (C.c_float * 8)()
        _chk(_lib().lf_last_phase_ms(self.h, out), "lf_last_phase_ms")
        return {_lib().lf_phase_name(i).decode(): float(out[i]) for i in range(8)}

    def fold_paths(self):
        m = C.c_uint()
        _chk(_lib().lf_last_fold_paths(self.h, C.byref(m)), "lf_last_fold_paths")
        return m.value

    def lin_split_rounds(self):
        """rounds of the last linearization sumcheck that ran in the split eq form -- test hook"""
        m = C.c_uint()
        _chk(_lib().lf_last_lin_split_rounds(self.h, C.byref(m)), "lf_last_lin_split_rounds")
        return m.value

    def fold_split_rounds(self):
        """mask (bit i-1 = round i) of the table rounds of the last folding sumcheck that ran in the split eq form -- test hook"""
        m = C.c_uint()
        _chk(_lib().lf_last_fold_split_rounds(self.h, C.byref(m)), "lf_last_fold_split_rounds")
        return m.value

    def timeline(self):
        """[(mark, ms since the start of the step)] of the last fold step (wall clock of the calling thread)"""
        names, ms = C.create_string_buffer(32 * 64), (C.c_double * 64)()
        n = _lib().lf_last_timeline(self.h, names, ms, 64)
        if n < 0:
            raise LfError(n, "lf_last_timeline")
        return [(names.raw[32 * i:32 * i + 32].split(b"\0")[0].decode(), float(ms[i])) for i in range(n)]

    def device_memory(self):
        """(free, total) bytes of the device (lf_device_memory)"""
        f, t = C.c_size_t(), C.c_size_t()
        _chk(_lib().lf_device_memory(self.h, C.byref(f), C.byref(t)), "lf_device_memory")
        return f.value, t.value

    def kernel_stats(self):
        f, a = C.c_float(), C.c_float()
        fn, an = C.c_int(), C.c_int()
        _chk(_lib().lf_last_kernel_stats(self.h, C.byref(f), C.byref(fn), C.byref(a), C.byref(an)), "lf_last_kernel_stats")
        return {"fold_round_ms": f.value, "fold_round_launches": fn.value, "ajtai_ms": a.value, "ajtai_launches": an.value}


class AjtaiCommitmentScheme:
    """commitment/commitment_scheme.rs:17-114.  The matrix lives on the device."""

    def __init__(self, ctx, matrix=None, kappa=None, n=None, seed=None):
        self.ctx = ctx
        if matrix is not None:  # AjtaiCommitmentScheme::new
            a, p = _a64(matrix)
            self._kappa, self._n = a.shape[0], a.shape[1]
            _chk(_lib().lf_ajtai_load(ctx.h, p, self._kappa, self._n), "lf_ajtai_load")
        else:                    # synthetic i.i.d. matrix generated on the device (bench)
            self._kappa, self._n = kappa, n
            _chk(_lib().lf_ajtai_generate(ctx.h, seed, kappa, n), "lf_ajtai_generate")

    def kappa(self):
        return self._kappa

    def width(self):
        return self._n

    def commit_ntt(self, f):
        """commit / commit_ntt: f is (n,d) or (batch,n,d)."""
        a, p = _a64(f)
        batch = 1 if a.ndim == 2 else a.shape[0]
        n = a.shape[-2]
        o = np.zeros((batch, self._kappa, self.ctx.RE), dtype=np.uint64)
        rc = _lib().lf_ajtai_commit(self.ctx.h, p, n, batch, o.ctypes.data_as(u64p))
        if rc == -1:
            raise CommitmentError(rc, f"WrongWitnessLength({n}, {self._n})")
        _chk(rc, "lf_ajtai_commit")
        return o[0] if a.ndim == 2 else o

    commit = commit_ntt


class PendingWitness:
    """a witness being ingested on the context's lowest-priority stream (lf_witness_from_w_ccs_begin / lf_witness_job_finish); keeps the host words alive until
    the job has finished"""

    def __init__(self, ctx, w_ccs):
        self.ctx = ctx
        self._keep, p = _a64(w_ccs)
        self._job = C.c_void_p()
        L = _lib()
        L.lf_witness_from_w_ccs_begin.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.lf_witness_job_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        _chk(L.lf_witness_from_w_ccs_begin(ctx.h, p, C.byref(self._job)), "lf_witness_from_w_ccs_begin")

    def result(self):
        if not self._job:
            raise RuntimeError("the job was finished already")
        h = C.c_void_p()
        job, self._job = self._job, None
        rc = _lib().lf_witness_job_finish(job, C.byref(h))
        self._keep = None
        _chk(rc, "lf_witness_job_finish")
        return Witness(self.ctx, h)

    def abandon(self):
        if self._job:
            job, self._job = self._job, None
            _lib().lf_witness_job_finish(job, None)
            self._keep = None

    def __del__(self):
        try:
            self.abandon()
        except Exception:
            pass


class Witness:
    """arith.rs:213-362; device-resident (centred f_coeff planes)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @classmethod
    def from_w_ccs(cls, ctx, w_ccs):
        a, p = _a64(w_ccs)
        h = C.c_void_p()
        _chk(_lib().lf_witness_from_w_ccs(ctx.h, p, C.byref(h)), "lf_witness_from_w_ccs")
        return cls(ctx, h)

    @classmethod
    def from_w_ccs_begin(cls, ctx, w_ccs):
        """lf_witness_from_w_ccs_begin: start building the witness next to whatever the context runs (a fold step); `.result()` waits and returns the Witness"""
        return PendingWitness(ctx, w_ccs)

    @classmethod
    def from_f_coeff(cls, ctx, f_coeff):
        a, p = _a64(f_coeff)
        h = C.c_void_p()
        _chk(_lib().lf_witness_from_f_coeff(ctx.h, p, C.byref(h)), "lf_witness_from_f_coeff")
        return cls(ctx, h)

    @classmethod
    def from_f(cls, ctx, f_ntt):
        a, p = _a64(f_ntt)
        h = C.c_void_p()
        _chk(_lib().lf_witness_from_f(ctx.h, p, C.byref(h)), "lf_witness_from_f")
        return cls(ctx, h)

    def _get(self, fn, count):
        o = np.zeros((count, self.ctx.RE), dtype=np.uint64)
        _chk(fn(self.ctx.h, self.h, o.ctypes.data_as(u64p)), "lf_witness_get")
        return o

    @property
    def f_coeff(self):
        return self._get(_lib().lf_witness_get_f_coeff, self.ctx.N)

    @property
    def f(self):
        return self._get(_lib().lf_witness_get_f, self.ctx.N)

    @property
    def w_ccs(self):
        return self._get(_lib().lf_witness_get_w_ccs, self.ctx.params.wit_len)

    def commit(self, scheme):
        o = np.zeros((scheme.kappa(), self.ctx.RE), dtype=np.uint64)
        _chk(_lib().lf_witness_commit(self.ctx.h, self.h, o.ctypes.data_as(u64p)), "lf_witness_commit")
        return o

    def free(self):
        if self.h:
            _lib().lf_witness_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PoseidonTranscript:
    """transcript/poseidon.rs:17-75 (host)."""

    def __init__(self, handle=None, ring="goldilocks"):
        self.ring = ring
        self.RE, self.TAU = RING_WORDS[ring], RING_TAU[ring]
        self.h = handle or C.c_void_p(_lib().lf_transcript_new_ring(RING_IDS[ring]))

    def clone(self):
        return PoseidonTranscript(C.c_void_p(_lib().lf_transcript_clone(self.h)), ring=self.ring)

    def absorb_fq(self, xs):
        a, p = _a64(xs)
        _lib().lf_transcript_absorb_fq(self.h, p, a.size)

    def absorb_slice(self, elems):
        a, p = _a64(elems)
        _lib().lf_transcript_absorb_ring(self.h, p, a.size // self.RE)

    absorb = absorb_slice

    def get_challenge(self):
        o = np.zeros(self.TAU, dtype=np.uint64)
        _lib().lf_transcript_get_challenge(self.h, o.ctypes.data_as(u64p))
        return o

    def get_short_challenge(self):
        o = np.zeros(self.RE, dtype=np.uint64)
        _lib().lf_transcript_get_short_challenge(self.h, o.ctypes.data_as(u64p))
        return o

    def __del__(self):
        try:
            if self.h:
                _lib().lf_transcript_free(self.h)
                self.h = None
        except Exception:
            pass


def poseidon_params(ring="goldilocks"):
    ark = np.zeros(720, dtype=np.uint64)
    mds = np.zeros(576, dtype=np.uint64)
    _lib().lf_poseidon_params_ring(ark.ctypes.data_as(u64p), mds.ctypes.data_as(u64p), RING_IDS[ring])
    return ark, mds


def poseidon_permute(state, plain=False, ring="goldilocks"):
    a = np.ascontiguousarray(state, dtype=np.uint64).copy()
    _lib().lf_poseidon_permute_ring(a.ctypes.data_as(u64p), int(plain), RING_IDS[ring])
    return a


class LFLinearizationProver:
    @staticmethod
    def prove(ctx, cm_i, wit, transcript):
        """nifs/linearization.rs:145-189 -> (lcccs flat, linearization proof flat)."""
        a, p = _a64(cm_i)
        lc = np.zeros((ctx.lcccs_len, ctx.RE), dtype=np.uint64)
        prm = ctx.params
        pr = np.zeros((prm.s * (prm.d + 2) + ctx.TAU + prm.t, ctx.RE), dtype=np.uint64)
        _chk(_lib().lf_linearize(ctx.h, transcript.h, p, wit.h, lc.ctypes.data_as(u64p), pr.ctypes.data_as(u64p)), "lf_linearize")
        return lc, pr


class LFDecompositionProver:
    @staticmethod
    def prove(ctx, lcccs, wit, transcript):
        """nifs/decomposition.rs:33-88 -> (K decomposed LCCCS flat [K*lcccs_len], decomposition proof flat).  The K decomposed
        witnesses are the base-b parts of `wit` and stay virtual on the device."""
        a, p = _a64(lcccs)
        prm = ctx.params
        lcs = np.zeros((prm.K * ctx.lcccs_len, ctx.RE), dtype=np.uint64)
        pr = np.zeros((prm.K * (prm.t + ctx.TAU + prm.l + 1 + prm.kappa), ctx.RE), dtype=np.uint64)
        _chk(_lib().lf_decomposition_prove(ctx.h, transcript.h, p, wit.h, lcs.ctypes.data_as(u64p), pr.ctypes.data_as(u64p)),
             "lf_decomposition_prove")
        return lcs, pr


class LFFoldingProver:
    @staticmethod
    def prove(ctx, lcccs_s, w_left, w_right, transcript):
        """nifs/folding.rs:42-130: lcccs_s = the 2K decomposed LCCCS, w_left / w_right the witnesses they are parts of
        -> (folded LCCCS flat, folded Witness, folding proof flat)."""
        a, p = _a64(lcccs_s)
        prm = ctx.params
        lc = np.zeros((ctx.lcccs_len, ctx.RE), dtype=np.uint64)
        pr = np.zeros((prm.s * (2 * prm.b + 1) + 2 * prm.K * (ctx.TAU + prm.t), ctx.RE), dtype=np.uint64)
        h = C.c_void_p()
        _chk(_lib().lf_folding_prove(ctx.h, transcript.h, p, w_left.h, w_right.h, lc.ctypes.data_as(u64p), C.byref(h),
                                     pr.ctypes.data_as(u64p)), "lf_folding_prove")
        return lc, Witness(ctx, h), pr


class NIFSProver:
    @staticmethod
    def prove(ctx, acc, w_acc, cm_i, w_i, transcript):
        """nifs.rs:48-103 -> (folded LCCCS flat, folded Witness, LFProof flat).  `ccs` and `scheme` of the
        reference signature are the ones loaded into ctx (load_ccs / AjtaiCommitmentScheme)."""
        a, pa = _a64(acc)
        b, pb = _a64(cm_i)
        lc = np.zeros((ctx.lcccs_len, ctx.RE), dtype=np.uint64)
        pr = np.zeros((ctx.proof_len, ctx.RE), dtype=np.uint64)
        h = C.c_void_p()
        _chk(_lib().lf_fold_step(ctx.h, transcript.h, pa, w_acc.h, pb, w_i.h, lc.ctypes.data_as(u64p), C.byref(h),
                                 pr.ctypes.data_as(u64p)), "lf_fold_step")
        return lc, Witness(ctx, h), pr


class NIFSVerifier:
    @staticmethod
    def verify(wl, acc, cm_i, proof, transcript):
        """nifs.rs:117-163 on the host (no GPU): wl gives the CCS shape (params, S, c).  Returns (ok, folded LCCCS flat,
        failed_stage)."""
        ring = wl.ring
        rid, RE_ = RING_IDS[ring], RING_WORDS[ring]
        prm = Params(wl.s, wl.wit_len, wl.l, wl.L, wl.K, wl.b, wl.B, wl.kappa, wl.t, wl.q, wl.d)
        so = np.ascontiguousarray(wl.S_off, dtype=np.uint32)
        si = np.ascontiguousarray(wl.S_idx, dtype=np.uint32)
        cc, pc = _a64(np.ascontiguousarray(wl.c).reshape(-1))
        a, pa = _a64(acc)
        b, pb = _a64(cm_i)
        pr, pp = _a64(proof)
        lc = np.zeros((_lib().lf_lcccs_len_ring(C.byref(prm), rid), RE_), dtype=np.uint64)
        st = C.c_int(0)
        rc = _lib().lf_verify_host(rid, C.byref(prm), so.ctypes.data_as(u32p), si.ctypes.data_as(u32p), pc, transcript.h, pa, pb, pp,
                                   lc.ctypes.data_as(u64p), C.byref(st))
        if rc not in (0, -8):
            raise LfError(rc, "lf_verify_host")
        return rc == 0, lc, st.value


def _wl_params(wl):
    return Params(wl.s, wl.wit_len, wl.l, wl.L, wl.K, wl.b, wl.B, wl.kappa, wl.t, wl.q, wl.d)


def proof_to_bytes(wl, proof):
    """LFProof::serialize_with_mode(Compress::Yes) (nifs.rs:28-34) of a flat proof; layout notes in include/lfhip.h."""
    prm, rid = _wl_params(wl), RING_IDS[wl.ring]
    a, p = _a64(proof)
    n = _lib().lf_proof_wire_size(C.byref(prm), rid)
    if a.size != _lib().lf_proof_len_ring(C.byref(prm), rid) * RING_WORDS[wl.ring]:
        raise LfError(-1, "proof_to_bytes: wrong proof length")
    buf = (C.c_uint8 * n)()
    _chk(_lib().lf_proof_serialize(C.byref(prm), rid, p, buf, n), "lf_proof_serialize")
    return bytes(buf)


def proof_from_bytes(wl, data):
    """LFProof::deserialize_with_mode(Compress::Yes, Validate::Yes) -> flat proof (ring elements x words)."""
    prm, rid = _wl_params(wl), RING_IDS[wl.ring]
    out = np.zeros((_lib().lf_proof_len_ring(C.byref(prm), rid), RING_WORDS[wl.ring]), dtype=np.uint64)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if len(data) else (C.c_uint8 * 1)()
    _chk(_lib().lf_proof_deserialize(C.byref(prm), rid, buf, len(data), out.ctypes.data_as(u64p)), "lf_proof_deserialize")
    return out


class MLSumcheckLin:
    """utils/sumcheck.rs:53-80 split at the transcript, for the linearization-shaped polynomial (lf_sumcheck_lin_*)."""

    def __init__(self, ctx, tables, eq_point):
        self.ctx = ctx
        a, p = _a64(tables)
        b, q = _a64(eq_point)
        _chk(_lib().lf_sumcheck_lin_begin(ctx.h, p, q), "lf_sumcheck_lin_begin")

    def prove_round(self, r_prev=None):
        prm = self.ctx.params
        o = np.zeros((prm.d + 2, self.ctx.RE), dtype=np.uint64)
        if r_prev is None:
            rc = _lib().lf_sumcheck_lin_round(self.ctx.h, None, o.ctypes.data_as(u64p))
        else:
            a, p = _a64(r_prev)
            rc = _lib().lf_sumcheck_lin_round(self.ctx.h, p, o.ctypes.data_as(u64p))
        _chk(rc, "lf_sumcheck_lin_round")
        return o

    def end(self):
        _chk(_lib().lf_sumcheck_lin_end(self.ctx.h), "lf_sumcheck_lin_end")


class MLSumcheckFold:
    """utils/sumcheck.rs:53-80 split at the transcript, for the folding polynomial (nifs/folding/utils.rs:200-325):
    tables = [eq_L, G_L, eq_R, G_R, eq_beta, f-hat ...] ((5 + 2K*tau) x m ring elements), mu = 2K challenges (lf_sumcheck_fold_*)."""

    def __init__(self, ctx, tables, mu):
        self.ctx = ctx
        a, p = _a64(tables)
        b, q = _a64(mu)
        _chk(_lib().lf_sumcheck_fold_begin(ctx.h, p, q), "lf_sumcheck_fold_begin")

    def prove_round(self, r_prev=None):
        prm = self.ctx.params
        o = np.zeros((2 * prm.b + 1, self.ctx.RE), dtype=np.uint64)
        if r_prev is None:
            rc = _lib().lf_sumcheck_fold_round(self.ctx.h, None, o.ctypes.data_as(u64p))
        else:
            a, p = _a64(r_prev)
            rc = _lib().lf_sumcheck_fold_round(self.ctx.h, p, o.ctypes.data_as(u64p))
        _chk(rc, "lf_sumcheck_fold_round")
        return o

    def end(self):
        _chk(_lib().lf_sumcheck_fold_end(self.ctx.h), "lf_sumcheck_fold_end")


def lincomb(ctx, coef, tables):
    """compute_f_0 (nifs/folding.rs:258-268): sum_i coef_i (.) tables_i; coef (n, RE), tables (n, len, RE)"""
    t = np.ascontiguousarray(tables, dtype=np.uint64)
    n, ln = t.shape[0], t.shape[1]
    a, p = _a64(coef)
    o = np.zeros((ln, ctx.RE), dtype=np.uint64)
    _chk(_lib().lf_lincomb(ctx.h, p, t.ctypes.data_as(u64p), n, ln, o.ctypes.data_as(u64p)), "lf_lincomb")
    return o


def horner_combine(ctx, tables, challenges):
    """calculate_challenged_mz_mle (nifs/folding.rs:208-226): tables (groups, per_group, len, RE), challenges (groups, TAU)"""
    t = np.ascontiguousarray(tables, dtype=np.uint64)
    g, pg, ln = t.shape[0], t.shape[1], t.shape[2]
    a, p = _a64(challenges)
    o = np.zeros((ln, ctx.RE), dtype=np.uint64)
    _chk(_lib().lf_horner_combine(ctx.h, t.ctypes.data_as(u64p), g, pg, ln, p, o.ctypes.data_as(u64p)), "lf_horner_combine")
    return o


def device_sponge(ctx, ops):
    """PoseidonSponge on the device (lf_device_sponge): ops = list of ("absorb", words) / ("squeeze", n) on a fresh sponge.
    Returns (list of squeezed arrays, state[26])."""
    codes, words, nout = [], [], 0
    for kind, arg in ops:
        if kind == "absorb":
            a = np.ascontiguousarray(arg, dtype=np.uint64).reshape(-1)
            codes.append(a.size)
            words.append(a)
        else:
            codes.append((1 << 24) | int(arg))
            nout += int(arg)
    cw = np.array(codes, dtype=np.uint32)
    w = np.concatenate(words) if words else np.zeros(0, dtype=np.uint64)
    out = np.zeros(max(nout, 1), dtype=np.uint64)
    st = np.zeros(26, dtype=np.uint64)
    _chk(_lib().lf_device_sponge(ctx.h, cw.ctypes.data_as(u32p), cw.size, w.ctypes.data_as(u64p), w.size, out.ctypes.data_as(u64p), nout,
                                 st.ctypes.data_as(u64p)), "lf_device_sponge")
    res, o = [], 0
    for kind, arg in ops:
        if kind != "absorb":
            res.append(out[o:o + int(arg)].copy())
            o += int(arg)
    return res, st
