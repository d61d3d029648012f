"""LatticeFold+ slice: Python mirror of the reference interface over the C ABI include/lfplus.h (ctypes, liblfhip.so).

Reference (crates/latticefold-plus): `RgInstance::from_f(f, &A, &DecompParameters{b, k, l})` (src/rgchk.rs:260-331), the double
commitment benchmarked by benches/double_commitment.rs; `utils::tensor` / `tensor_product` (src/utils.rs:45-83).  Ring: FrogRing
RqPoly, Z_p[X]/(X^16 + 1) in coefficient form, 16 canonical u64 words per element.  GPU only: there is no CPU path."""
import ctypes as C
import math
import re
import os
from dataclasses import dataclass

import numpy as np

from . import api

P = 15912092521325583641
D = 16
u64p = C.POINTER(C.c_uint64)
i8p = C.POINTER(C.c_int8)
_READY = False


class LfPlusError(RuntimeError):
    def __init__(self, code, msg):
        self.code = code
        super().__init__(f"liblfhip (lfplus): {msg} ({code})")


E_ARG, E_NO_DEVICE, E_HIP, E_EXP_DOMAIN, E_SMALL_N, E_REJECT = -1, -2, -3, -4, -5, -6
ABSENT = -128     # exponent digit of a zero entry of a monomial set (lfplus.h LFPLUS_ABSENT)


def exported_symbols():
    """every symbol include/lfplus.h declares"""
    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "lfplus.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lfplus_[a-z0-9_]+)\s*\(", hdr)))


def _lib():
    global _READY
    L = api._lib()
    if not _READY:
        vp = C.c_void_p
        L.lfplus_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.lfplus_ctx_destroy.argtypes = [vp]
        L.lfplus_ctx_destroy.restype = None
        L.lfplus_scratch_trim.argtypes = [C.c_int]
        L.lfplus_scratch_trim.restype = None
        L.lfplus_scratch_bytes.argtypes = [C.c_int]
        L.lfplus_scratch_bytes.restype = C.c_size_t
        L.lfplus_last_error.argtypes = [vp]
        L.lfplus_last_error.restype = C.c_char_p
        L.lfplus_set_matrix.argtypes = [vp, u64p, C.c_uint32, C.c_uint64]
        L.lfplus_set_witness.argtypes = [vp, u64p, C.c_uint64]
        L.lfplus_rg_from_f.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32]
        L.lfplus_rg_from_f_async.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32]
        L.lfplus_join_async.argtypes = [vp]
        L.lfplus_rg_read.argtypes = [vp, i8p, u64p, u64p, i8p, u64p, u64p, u64p]
        L.lfplus_rg_from_f_timed.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        L.lfplus_commit.argtypes = [vp, u64p, C.c_uint64, u64p]
        u32pp, u64pp = C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64p)
        L.lfplus_decompose.argtypes = [vp, C.c_uint64, u64p, u64p, C.c_uint32, u32pp, u32pp, u64pp, u64p, u64p, u64p, u64p, u64p, u64p]
        L.lfplus_decompose_resident.argtypes = [vp, C.c_uint64, u64p, u64p, C.c_uint32, u32pp, u32pp, u64pp, vp, vp, u64p, u64p, u64p, u64p]
        L.lfplus_get_witness.argtypes = [vp, u64p, C.c_uint64]
        L.lfplus_tensor.argtypes = [vp, u64p, C.c_uint32, u64p]
        L.lfplus_tensor_product.argtypes = [vp, u64p, C.c_uint64, u64p, C.c_uint64, u64p]
        u8p, vpp, ip = C.POINTER(C.c_uint8), C.POINTER(vp), C.POINTER(C.c_int)
        L.lfplus_transcript_new.restype = vp
        L.lfplus_transcript_clone.restype = vp
        L.lfplus_transcript_clone.argtypes = [vp]
        L.lfplus_transcript_free.argtypes = [vp]
        L.lfplus_transcript_free.restype = None
        L.lfplus_transcript_absorb.argtypes = [vp, u64p, C.c_size_t]
        L.lfplus_transcript_challenge.argtypes = [vp, u64p]
        L.lfplus_transcript_squeeze_bytes.argtypes = [vp, C.c_size_t, u8p]
        L.lfplus_short_challenge.argtypes = [vp, u64p]
        L.lfplus_poseidon_params.argtypes = [u64p, u64p]
        L.lfplus_poseidon_permute.argtypes = [u64p, C.c_int]
        L.lfplus_set_check.argtypes = [vp, vp, C.c_uint32, i8p, C.c_uint32, C.c_uint32, i8p, C.c_uint32, C.c_uint32, u32pp, u32pp, u64pp, u64p, u64p, u64p, u64p]
        L.lfplus_set_check_verify.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u64p, u64p, u64p, u64p, ip]
        L.lfplus_range_check.argtypes = [vpp, C.c_uint32, vp, C.c_uint32, u32pp, u32pp, u64pp] + [u64p] * 8
        L.lfplus_range_check_verify.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32] + [u64p] * 8 + [ip]
        L.lfplus_cm_prove.argtypes = [vpp, C.c_uint32, vp, C.c_uint32, C.c_uint32, u32pp, u32pp, u64pp] + [u64p] * 17
        L.lfplus_cm_read_g.argtypes = [vp, u64p]
        L.lfplus_share_matrix.argtypes = [vp, vp]
        L.lfplus_set_matrices.argtypes = [vp, C.c_uint64, C.c_uint32, u32pp, u32pp, u64pp]
        L.lfplus_share_matrices.argtypes = [vp, vp]
        L.lfplus_r1cs_linearize.argtypes = [vp, vp, u32pp, u32pp, u64pp, u64p, u64p, u64p]
        L.lfplus_r1cs_verify.argtypes = [vp, C.c_uint32, u64p, u64p, u64p, ip]
        L.lfplus_decomp_verify.argtypes = [u64p, u64p, C.c_uint32, u64p, u64p, C.c_uint32, u64p, u64p, C.c_uint64, ip]
        L.lfplus_mlin.argtypes = [vpp, C.c_uint32, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, u32pp, u32pp, u64pp] + [u64p] * 19
        L.lfplus_cm_verify.argtypes = [vp] + [C.c_uint32] * 6 + [u64pp] + [u64p] * 15 + [ip]
        L.lfplus_set_sharding.argtypes = [vp, C.c_int, C.c_int, api.EXCHANGE_FN, vp]
        L.lfplus_dist_unique_id.argtypes = [u8p]
        L.lfplus_dist_init.argtypes = [vp, C.c_int, C.c_int, u8p]
        L.lfplus_dist_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
        _READY = True
    return L


def _w(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(u64p)


@dataclass
class DecompParameters:
    """rgchk.rs:20-24"""
    b: int
    k: int
    l: int

    @staticmethod
    def for_frog(k, b=D // 2):
        """l = ceil(log_{d/2} q) as every reference call site computes it (rgchk.rs:369-371, benches/double_commitment.rs:68-70)"""
        return DecompParameters(b, k, math.ceil(math.log(float(P)) / math.log(D / 2)))


def scratch_bytes(device=0):
    """Bytes of idle scratch the process-wide cache holds on `device` (lfplus_scratch_bytes)"""
    return int(_lib().lfplus_scratch_bytes(int(device)))


def scratch_trim(device=-1):
    """Frees the scratch blocks destroyed contexts left in the process-wide cache (lfplus_scratch_trim; device < 0: every device)"""
    _lib().lfplus_scratch_trim(int(device))


class PlusContext:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = _lib().lfplus_ctx_create(device, C.byref(self.h))
        if rc:
            raise LfPlusError(rc, "lfplus_ctx_create: no usable HIP device (the library has no CPU path)")
        self.kappa = self.n = 0

    def _chk(self, rc):
        if rc:
            raise LfPlusError(rc, _lib().lfplus_last_error(self.h).decode())

    def close(self):
        if self.h:
            _lib().lfplus_ctx_destroy(self.h)
            self.h = C.c_void_p()

    shard = (0, 1)     # (rank, world) of a column-sharded prover

    def set_sharding_model(self, rank, world):
        """lfplus_set_sharding_model: a TIMING model of rank `rank` of `world` with no peers (tools/shard_model.py --lfplus); what such a prover returns is not a proof"""
        L = _lib()
        L.lfplus_set_sharding_model.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._chk(L.lfplus_set_sharding_model(self.h, int(rank), int(world)))
        self.shard = (rank, world)

    def dist_stats_words(self, reset=False):
        L = _lib()
        L.lfplus_dist_stats_words.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        w = C.c_uint64()
        self._chk(L.lfplus_dist_stats_words(self.h, C.byref(w), int(reset)))
        return w.value

    def set_sharding(self, rank, world, allgather):
        """Column sharding over a HOST transport (lfplus_set_sharding); call before set_matrix.  allgather(np.uint64[words]) -> np.uint64[world, words]
        in rank order (latticefold_amd.dist.make_allgather)."""
        def _cb(user, send, recv, words):
            try:
                mine = np.ctypeslib.as_array(send, shape=(words,)).copy()
                out = np.ascontiguousarray(allgather(mine), dtype=np.uint64).reshape(world * words)
                C.memmove(recv, out.ctypes.data, world * words * 8)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                import sys
                print("lfplus exchange callback failed:", repr(e), file=sys.stderr)
                return -1
        self._exchange_cb = api.EXCHANGE_FN(_cb) if world > 1 else api.EXCHANGE_FN(0)
        self._chk(_lib().lfplus_set_sharding(self.h, rank, world, self._exchange_cb, None))
        self.shard = (rank, world)

    def dist_init(self, rank, world, id128):
        """Column sharding over RCCL (lfplus_dist_init): id128 = the bytes of an ncclUniqueId (plus.dist_unique_id() on rank 0, broadcast by the launcher)"""
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        self._chk(_lib().lfplus_dist_init(self.h, rank, world, buf))
        self.shard = (rank, world)

    def dist_stats(self, reset=False):
        n, tot, mx = C.c_uint64(), C.c_double(), C.c_double()
        self._chk(_lib().lfplus_dist_stats(self.h, C.byref(n), C.byref(tot), C.byref(mx), int(reset)))
        return n.value, tot.value, mx.value

    def set_matrix(self, A):
        """A: (kappa, n, 16) canonical words (Matrix<R>, coefficient form); in a sharded context the rank's columns (kappa, n / world, 16)"""
        A, p = _w(A)
        assert A.ndim == 3 and A.shape[2] == D
        self._chk(_lib().lfplus_set_matrix(self.h, p, A.shape[0], A.shape[1]))
        self.kappa, self.n = A.shape[0], A.shape[1] * self.shard[1]

    def share_matrix(self, other):
        """use `other`'s resident commitment matrix (no copy, reference-counted) -- and, for a sharded prover, its transport"""
        self._chk(_lib().lfplus_share_matrix(self.h, other.h))
        self.kappa, self.n, self.shard = other.kappa, other.n, other.shard

    def set_matrices(self, M, n=None):
        """make the constraint-system matrices resident (CSR triples); calls that get M = RESIDENT then use them without another upload"""
        keep, rp, cp, vp = _csr_args(M)
        self._chk(_lib().lfplus_set_matrices(self.h, n if n is not None else self.n, len(M), rp, cp, vp))
        self._nres = len(M)

    def share_matrices(self, other):
        self._chk(_lib().lfplus_share_matrices(self.h, other.h))

    def set_witness(self, f):
        f, p = _w(f)
        assert f.ndim == 2 and f.shape[1] == D
        self._chk(_lib().lfplus_set_witness(self.h, p, f.shape[0]))

    def commit(self, v):
        """Matrix::try_mul_vec"""
        v, p = _w(v)
        out = np.zeros((self.kappa, D), dtype=np.uint64)
        self._chk(_lib().lfplus_commit(self.h, p, v.shape[0], out.ctypes.data_as(u64p)))
        return out

    def tensor(self, r):
        r, p = _w([int(x) % P for x in r])
        out = np.zeros(1 << r.size, dtype=np.uint64)
        self._chk(_lib().lfplus_tensor(self.h, p, r.size, out.ctypes.data_as(u64p)))
        return out

    def tensor_product(self, a, b):
        a, pa = _w([int(x) % P for x in a])
        b, pb = _w([int(x) % P for x in b])
        out = np.zeros(a.size * b.size if a.size and b.size else a.size + b.size, dtype=np.uint64)
        self._chk(_lib().lfplus_tensor_product(self.h, pa, a.size, pb, b.size, out.ctypes.data_as(u64p)))
        return out

    def get_witness(self):
        """the context's resident witness (n, 16), read back from the device"""
        out = np.zeros((self.n, D), dtype=np.uint64)
        self._chk(_lib().lfplus_get_witness(self.h, out.ctypes.data_as(u64p), self.n))
        return out

    def decompose(self, f, A, B, r, M=(), into=None, out_bufs=None):
        """Decomp{f, r, M}.decompose(&A, B) (decomp.rs:32-99).  r: (nvars, 2, 16) pairs of ring elements; M: CSR matrices (rowptr, col, val[nnz][16]).
        -> dict(F0, F1 (n,16); C0, C1 (kappa,16); v0, v1 (1+len(M), 2, 16)): ((LinB0, LinB1), DecompProof) of the reference, flat.
        into = (ctx0, ctx1): F0 / F1 stay on the device as the resident witnesses of those contexts (lfplus_decompose_resident) and are not returned;
        out_bufs = (F0, F1): host arrays to receive them (already touched: a download into fresh pages is page-fault-bound, 10 ms instead of 2.4 per 2^20 rows)"""
        if A is not None:
            self.set_matrix(A)
        if f is not None:              # None: the resident witness (e.g. the folded g Mlin.mlin left there)
            self.set_witness(f)
        r = np.ascontiguousarray(r, dtype=np.uint64)
        r_a, r_b = np.ascontiguousarray(r[:, 0]), np.ascontiguousarray(r[:, 1])
        nm = len(M)
        keep, rp, cp, vp_ = _csr_args(M)
        n, kappa = self.n, self.kappa
        out = {"C0": np.zeros((kappa, D), dtype=np.uint64), "C1": np.zeros((kappa, D), dtype=np.uint64), "v0": np.zeros((1 + nm, 2, D), dtype=np.uint64),
               "v1": np.zeros((1 + nm, 2, D), dtype=np.uint64)}
        if into is not None:
            self._chk(_lib().lfplus_decompose_resident(self.h, B, r_a.ctypes.data_as(u64p), r_b.ctypes.data_as(u64p), nm, rp, cp, vp_, into[0].h, into[1].h,
                                                       *[out[k].ctypes.data_as(u64p) for k in ("C0", "C1", "v0", "v1")]))
            return out
        if out_bufs is not None and all(isinstance(b, np.ndarray) and b.shape == (n, D) and b.dtype == np.uint64 and b.flags.c_contiguous for b in out_bufs):
            out["F0"], out["F1"] = out_bufs
        else:
            out["F0"], out["F1"] = np.zeros((n, D), dtype=np.uint64), np.zeros((n, D), dtype=np.uint64)
        self._chk(_lib().lfplus_decompose(self.h, B, r_a.ctypes.data_as(u64p), r_b.ctypes.data_as(u64p), nm, rp, cp, vp_,
                                          *[out[k].ctypes.data_as(u64p) for k in ("F0", "F1", "C0", "C1", "v0", "v1")]))
        return out

    def rg_from_f_async(self, dparams):
        """lfplus_rg_from_f_async: RgInstance::from_f of the resident witness on the context's second stream (collected by rg_from_f / mlin with the same parameters)"""
        self._chk(_lib().lfplus_rg_from_f_async(self.h, dparams.b, dparams.k, dparams.l))

    def join_async(self):
        self._chk(_lib().lfplus_join_async(self.h))

    def time_rg_from_f(self, dparams, iters):
        ms = C.c_double()
        self._chk(_lib().lfplus_rg_from_f_timed(self.h, dparams.b, dparams.k, dparams.l, iters, C.byref(ms)))
        return ms.value


@dataclass
class FComs:
    """rgchk.rs:26-31"""
    cm_f: np.ndarray
    C_Mf: np.ndarray
    cm_mtau: np.ndarray


@dataclass
class RgInstance:
    """rgchk.rs:40-48.  M_f / m_tau are unit monomials, held as their exponent digits D_f / centred tau (exp(a) = X^a, X^(d+a) for a < 0)."""
    D_f: np.ndarray      # (k, n, 16) int8
    tau: np.ndarray      # (n,)
    m_tau_exp: np.ndarray  # (n,) int8
    f: np.ndarray
    comM_f: np.ndarray   # (k, kappa, 16, 16): comM_f[k_i] is a kappa x d matrix of ring elements
    fcoms: FComs

    @staticmethod
    def from_f(ctx, f, A, dparams):
        """RgInstance::from_f(f, &A, &decomp) (rgchk.rs:260-331).  A = None keeps the matrix already resident in ctx."""
        if A is not None:
            ctx.set_matrix(A)
        ctx.set_witness(f)
        ctx._chk(_lib().lfplus_rg_from_f(ctx.h, dparams.b, dparams.k, dparams.l))
        ctx._k = dparams.k
        k, n, kappa = dparams.k, ctx.n, ctx.kappa
        Df = np.zeros((k, n, D), dtype=np.int8)
        com = np.zeros((k, kappa, D, D), dtype=np.uint64)
        tau = np.zeros(n, dtype=np.uint64)
        mt = np.zeros(n, dtype=np.int8)
        c = [np.zeros((kappa, D), dtype=np.uint64) for _ in range(3)]
        ctx._chk(_lib().lfplus_rg_read(ctx.h, Df.ctypes.data_as(i8p), com.ctypes.data_as(u64p), tau.ctypes.data_as(u64p), mt.ctypes.data_as(i8p),
                                       *[x.ctypes.data_as(u64p) for x in c]))
        return RgInstance(Df, tau, mt, np.asarray(f), com, FComs(*c))

    def M_f(self, ki, rows=slice(None)):
        """dense monomial matrix exp(D_f[ki]) for the given rows: (rows, 16 columns, 16 words)"""
        return exp(self.D_f[ki, rows])


def dist_unique_id():
    """an ncclUniqueId (128 bytes) for PlusContext.dist_init"""
    buf = (C.c_uint8 * 128)()
    rc = _lib().lfplus_dist_unique_id(buf)
    if rc:
        raise LfPlusError(rc, "lfplus_dist_unique_id: RCCL not loadable")
    return bytes(buf)


def exp(digits):
    """stark_rings exp: digit a in (-d/2, d/2) -> the unit monomial X^a (a >= 0) / X^(d + a) (a < 0), as 16-word elements"""
    dg = np.asarray(digits, dtype=np.int64)
    if (np.abs(dg) >= D // 2).any():
        raise ValueError("exp: digit outside (-d/2, d/2)")
    out = np.zeros(dg.shape + (D,), dtype=np.uint64)
    np.put_along_axis(out, np.where(dg >= 0, dg, D + dg)[..., None], 1, axis=-1)
    return out


# ---- the transcript-driven part (src/transcript.rs, setchk.rs, rgchk.rs:81-258) ------------------------------------------------------------
class _Resident(tuple):
    """M = RESIDENT(count): use the matrices lfplus_set_matrices left in the (first) context"""


def RESIDENT(count):
    return _Resident((None,) * count)


def _csr_args(mats):
    """mats: list of (rowptr uint32 [n+1], col uint32 [nnz], val uint64 [nnz][16]), or RESIDENT(count)"""
    if isinstance(mats, _Resident):
        return [], None, None, None
    keep = [(np.ascontiguousarray(r, dtype=np.uint32), np.ascontiguousarray(c, dtype=np.uint32), np.ascontiguousarray(v, dtype=np.uint64)) for r, c, v in mats]
    u32p = C.POINTER(C.c_uint32)
    n = max(1, len(keep))
    return keep, (u32p * n)(*[k[0].ctypes.data_as(u32p) for k in keep]), (u32p * n)(*[k[1].ctypes.data_as(u32p) for k in keep]), \
        (u64p * n)(*[k[2].ctypes.data_as(u64p) for k in keep])


class PoseidonTranscript:
    """PoseidonTranscript::<RqPoly>::empty::<FrogPoseidonConfig>() (src/transcript.rs:20-78); host object, no GPU needed"""

    def __init__(self, h=None):
        self.h = h if h is not None else _lib().lfplus_transcript_new()

    def clone(self):
        return PoseidonTranscript(_lib().lfplus_transcript_clone(self.h))

    def absorb(self, ring):
        a = np.ascontiguousarray(ring, dtype=np.uint64).reshape(-1, D)
        _lib().lfplus_transcript_absorb(self.h, a.ctypes.data_as(u64p), a.shape[0])

    def get_challenge(self):
        o = C.c_uint64()
        _lib().lfplus_transcript_challenge(self.h, C.byref(o))
        return o.value

    def squeeze_bytes(self, n):
        o = np.zeros(n, dtype=np.uint8)
        _lib().lfplus_transcript_squeeze_bytes(self.h, n, o.ctypes.data_as(C.POINTER(C.c_uint8)))
        return o

    def short_challenge(self):
        """utils::short_challenge(128, transcript) (src/utils.rs:87-101)"""
        o = np.zeros(D, dtype=np.uint64)
        _lib().lfplus_short_challenge(self.h, o.ctypes.data_as(u64p))
        return o

    def __del__(self):
        try:
            _lib().lfplus_transcript_free(self.h)
        except Exception:
            pass


def poseidon_simd():
    """True when the transcript's permutation runs on the host's AVX-512 IFMA lanes (lfp_poseidon_simd.cc)"""
    L = _lib()
    L.lfplus_poseidon_simd.restype = C.c_int
    return bool(L.lfplus_poseidon_simd())


def poseidon_permute(state, plain=False):
    """one Poseidon permutation of 24 canonical words; plain = True / 1: the textbook definition, 2: the scalar sparse form, False / 0: the form the
    transcript runs (AVX-512 IFMA lanes when the CPU has them)"""
    st = np.ascontiguousarray(state, dtype=np.uint64).copy()
    assert st.shape == (24,)
    rc = _lib().lfplus_poseidon_permute(st.ctypes.data_as(u64p), int(plain))
    if rc:
        raise LfPlusError(rc, "lfplus_poseidon_permute")
    return st


def poseidon_params():
    ark, mds = np.zeros(720, dtype=np.uint64), np.zeros(576, dtype=np.uint64)
    _lib().lfplus_poseidon_params(ark.ctypes.data_as(u64p), mds.ctypes.data_as(u64p))
    return ark, mds


def set_check(ctx, transcript, nvars, mat_digits, vec_digits=None, M=()):
    """In::set_check (src/setchk.rs:65-262).  mat_digits (nmat, n, ncols) / vec_digits (nvec, n): exponent digits (int8; ABSENT = zero entry)
    -> dict(r, msgs (nvars, 4, 16), e (1 + len(M), nmat, ncols, 16), b (nvec, 16))"""
    md = np.ascontiguousarray(mat_digits, dtype=np.int8)
    nmat, n, ncols = md.shape
    vd = np.zeros((0, n), dtype=np.int8) if vec_digits is None else np.ascontiguousarray(vec_digits, dtype=np.int8)
    nvec, nM = vd.shape[0], len(M)
    keep, rp, cp, vp = _csr_args(M)
    r, msgs = np.zeros(nvars, dtype=np.uint64), np.zeros((nvars, 4, D), dtype=np.uint64)
    e, b = np.zeros((1 + nM, nmat, ncols, D), dtype=np.uint64), np.zeros((max(nvec, 1), D), dtype=np.uint64)
    ctx._chk(_lib().lfplus_set_check(ctx.h, transcript.h, nvars, md.ctypes.data_as(i8p), nmat, ncols, vd.ctypes.data_as(i8p) if nvec else None, nvec, nM, rp, cp, vp,
                                     r.ctypes.data_as(u64p), msgs.ctypes.data_as(u64p), e.ctypes.data_as(u64p), b.ctypes.data_as(u64p)))
    return {"r": r, "msgs": msgs, "e": e, "b": b[:nvec]}


def _shaped(proof, key, shape):
    """a proof array as contiguous u64 words of exactly `shape` -- the C verifiers index raw pointers with the VERIFIER's parameters, so a proof whose arrays
    are shorter than those parameters imply must be refused here (LFPLUS_E_ARG), never read"""
    try:
        a = np.ascontiguousarray(proof[key], dtype=np.uint64)
    except (KeyError, TypeError, ValueError, OverflowError) as ex:
        raise LfPlusError(E_ARG, f"malformed proof: field {key!r}: {ex}")
    if a.shape != tuple(shape):
        raise LfPlusError(E_ARG, f"malformed proof: field {key!r} has shape {a.shape}, the verifier's parameters require {tuple(shape)}")
    return a


def _nvars_ok(nvars):
    if not isinstance(nvars, (int, np.integer)) or not 1 <= int(nvars) <= 32:
        raise LfPlusError(E_ARG, "nvars outside [1, 32]")
    return int(nvars)


def set_check_verify(transcript, nvars, out, nM=0, nmat=None, ncols=None, nvec=None):
    """Out::verify (src/setchk.rs:266-340) on the host -> (accepted, stage, r).  nvars and nM are the VERIFIER's; nmat / ncols / nvec default to the shape of
    out["e"] / out["b"], and every array is checked against them before the C verifier sees a pointer"""
    nvars = _nvars_ok(nvars)
    e0, b0 = np.asarray(out["e"]), np.asarray(out["b"])
    if e0.ndim != 4 or b0.ndim != 2:
        raise LfPlusError(E_ARG, "malformed proof: e must be (1 + nM, nmat, ncols, 16), b (nvec, 16)")
    nmat, ncols, nvec = (e0.shape[1] if nmat is None else nmat), (e0.shape[2] if ncols is None else ncols), (b0.shape[0] if nvec is None else nvec)
    if nmat < 1 or ncols < 1:
        raise LfPlusError(E_ARG, "malformed proof: no matrix set")
    e, b, msgs = _shaped(out, "e", (1 + nM, nmat, ncols, D)), _shaped(out, "b", (nvec, D)), _shaped(out, "msgs", (nvars, 4, D))
    bb = b if nvec else np.zeros((1, D), dtype=np.uint64)
    r, st = np.zeros(nvars, dtype=np.uint64), C.c_int()
    rc = _lib().lfplus_set_check_verify(transcript.h, nvars, nmat, ncols, nvec, nM, msgs.ctypes.data_as(u64p), e.ctypes.data_as(u64p), bb.ctypes.data_as(u64p),
                                        r.ctypes.data_as(u64p), C.byref(st))
    if rc not in (0, E_REJECT):
        raise LfPlusError(rc, "lfplus_set_check_verify")
    return rc == 0, st.value, r


def range_check(ctxs, transcript, M=()):
    """Rg::range_check (src/rgchk.rs:81-186) over the resident RgInstances of `ctxs` (each after RgInstance.from_f) -> dict of the Dcom fields"""
    L, nM, c0 = len(ctxs), len(M), ctxs[0]
    if not c0.n or not getattr(c0, "_k", 0):
        raise LfPlusError(E_ARG, "range_check: no resident RgInstance (run RgInstance.from_f first)")
    n, k = c0.n, c0._k
    nvars = n.bit_length() - 1
    keep, rp, cp, vp = _csr_args(M)
    hs = (C.c_void_p * L)(*[c.h for c in ctxs])
    r, msgs = np.zeros(nvars, dtype=np.uint64), np.zeros((nvars, 4, D), dtype=np.uint64)
    e, b = np.zeros((1 + nM, L * k, D, D), dtype=np.uint64), np.zeros((L, D), dtype=np.uint64)
    v, a = np.zeros((L, D), dtype=np.uint64), np.zeros((L, 1 + nM), dtype=np.uint64)
    bb, c = np.zeros((L, 1 + nM, D), dtype=np.uint64), np.zeros((L, 1 + nM, D), dtype=np.uint64)
    c0._chk(_lib().lfplus_range_check(hs, L, transcript.h, nM, rp, cp, vp, *[x.ctypes.data_as(u64p) for x in (r, msgs, e, b, v, a, bb, c)]))
    return {"r": r, "msgs": msgs, "e": e, "b": b, "v": v, "a": a, "bb": bb, "c": c, "k": k, "nvars": nvars}


def _dcom_arrays(d, nvars, L, k, nM):
    """the Dcom fields (rgchk.rs:50-79) checked against the verifier's nvars, L, k, nM"""
    return {"msgs": _shaped(d, "msgs", (nvars, 4, D)), "e": _shaped(d, "e", (1 + nM, L * k, D, D)), "b": _shaped(d, "b", (L, D)), "v": _shaped(d, "v", (L, D)),
            "a": _shaped(d, "a", (L, 1 + nM)), "bb": _shaped(d, "bb", (L, 1 + nM, D)), "c": _shaped(d, "c", (L, 1 + nM, D))}


def range_check_verify(transcript, d, nvars=None, L=None, k=None, nM=None):
    """Dcom::verify (src/rgchk.rs:193-258) on the host -> (accepted, stage, r).  nvars / L / k / nM are the VERIFIER's parameters (a caller without its own
    takes them from the proof's metadata -- fine for a self-check, not for an untrusted proof); every array is checked against them"""
    nvars = _nvars_ok(d["nvars"] if nvars is None else nvars)
    k = int(d["k"] if k is None else k)
    L = int(np.asarray(d["b"]).shape[0] if L is None else L)
    nM = int(np.asarray(d["a"]).shape[-1] - 1 if nM is None else nM)
    if L < 1 or k < 1 or nM < 0:
        raise LfPlusError(E_ARG, "range_check_verify: bad parameters")
    arr = _dcom_arrays(d, nvars, L, k, nM)
    r, st = np.zeros(nvars, dtype=np.uint64), C.c_int()
    rc = _lib().lfplus_range_check_verify(transcript.h, nvars, L, k, nM, *[arr[key].ctypes.data_as(u64p) for key in ("msgs", "e", "b", "v", "a", "bb", "c")],
                                          r.ctypes.data_as(u64p), C.byref(st))
    if rc not in (0, E_REJECT):
        raise LfPlusError(rc, "lfplus_range_check_verify")
    return rc == 0, st.value, r


def cm_prove(ctxs, transcript, ell, M=(), want_g=False):
    """Cm::prove (src/cm.rs:56-347) over the resident RgInstances of `ctxs` -> dict of the CmProof fields (dcom = the range check's fields, comh,
    sumcheck proofs pa / pb, evals ea / eb) and of the folded instance x = (cm_g, ro, vo); `g` (L, n, 16) when want_g, else it stays on the device"""
    L, nM, c0 = len(ctxs), len(M), ctxs[0] if ctxs else None
    if not ctxs or not c0.n or not getattr(c0, "_k", 0):
        raise LfPlusError(E_ARG, "cm_prove: no resident RgInstance (run RgInstance.from_f first)")
    n, k, kappa = c0.n, c0._k, c0.kappa
    nvars = n.bit_length() - 1
    per = 4 + 4 * nM
    keep, rp, cp, vp = _csr_args(M)
    hs = (C.c_void_p * L)(*[c.h for c in ctxs])
    z = lambda *shape: np.zeros(shape, dtype=np.uint64)
    o = {"r": z(nvars), "msgs": z(nvars, 4, D), "e": z(1 + nM, L * k, D, D), "b": z(L, D), "v": z(L, D), "a": z(L, 1 + nM), "bb": z(L, 1 + nM, D),
         "c": z(L, 1 + nM, D), "comh": z(L, kappa, D), "pa": z(nvars, 3, D), "pb": z(nvars, 3, D), "ea": z(L, per, D), "eb": z(L, per, D),
         "cm_g": z(L, kappa, D), "ro": z(2, nvars), "vo": z(L, 1 + nM, 2, D)}
    g = z(L, n, D) if want_g else None
    keys = ("r", "msgs", "e", "b", "v", "a", "bb", "c", "comh", "pa", "pb", "ea", "eb", "cm_g", "ro", "vo")
    c0._chk(_lib().lfplus_cm_prove(hs, L, transcript.h, ell, nM, rp, cp, vp, *[o[key].ctypes.data_as(u64p) for key in keys],
                                   g.ctypes.data_as(u64p) if want_g else None))
    o.update(k=k, ell=ell, kappa=kappa, nvars=nvars)
    if want_g:
        o["g"] = g
    return o


def cm_read_g(ctx):
    """the folded witness g of the last cm_prove this context took part in: (n, 16) canonical words"""
    g = np.zeros((ctx.n, D), dtype=np.uint64)
    ctx._chk(_lib().lfplus_cm_read_g(ctx.h, g.ctypes.data_as(u64p)))
    return g


def cm_verify(transcript, proof, fcoms, nvars=None, L=None, k=None, ell=None, kappa=None, nM=None):
    """CmProof::verify (src/cm.rs:349-543) on the host.  fcoms[l] = (3, kappa, 16): cm_f | C_Mf | cm_mtau -> (accepted, stage, dict(cm_g, ro, vo)).
    nvars .. nM are the VERIFIER's parameters (CmProof::verify takes M.len() and the parameters from its own side, cm.rs:349-365); left None they come from
    the proof's metadata, which only a self-check should do.  Every array is checked against them before the C verifier sees a pointer"""
    nvars = _nvars_ok(proof["nvars"] if nvars is None else nvars)
    k, ell, kappa = (int(proof[key] if val is None else val) for key, val in (("k", k), ("ell", ell), ("kappa", kappa)))
    L = int(np.asarray(proof["b"]).shape[0] if L is None else L)
    nM = int(np.asarray(proof["a"]).shape[-1] - 1 if nM is None else nM)
    if L < 1 or not 1 <= k <= 16 or not 1 <= ell <= 64 or not 1 <= kappa <= 64 or not 0 <= nM <= 64:
        raise LfPlusError(E_ARG, "cm_verify: parameters outside the envelope")
    per = 4 + 4 * nM
    keys = ("msgs", "e", "b", "v", "a", "bb", "c", "comh", "pa", "pb", "ea", "eb")
    arr = _dcom_arrays(proof, nvars, L, k, nM)
    arr.update(comh=_shaped(proof, "comh", (L, kappa, D)), pa=_shaped(proof, "pa", (nvars, 3, D)), pb=_shaped(proof, "pb", (nvars, 3, D)),
               ea=_shaped(proof, "ea", (L, per, D)), eb=_shaped(proof, "eb", (L, per, D)))
    fc = [np.ascontiguousarray(x, dtype=np.uint64) for x in fcoms]
    if len(fc) != L or any(x.shape != (3, kappa, D) for x in fc):
        raise LfPlusError(E_ARG, f"cm_verify: fcoms must be {L} arrays of shape (3, {kappa}, 16)")
    fptr = (u64p * L)(*[x.ctypes.data_as(u64p) for x in fc])
    x = {"cm_g": np.zeros((L, kappa, D), dtype=np.uint64), "ro": np.zeros((2, nvars), dtype=np.uint64), "vo": np.zeros((L, 1 + nM, 2, D), dtype=np.uint64)}
    st = C.c_int()
    rc = _lib().lfplus_cm_verify(transcript.h, nvars, L, k, ell, kappa, nM, fptr, *[arr[key].ctypes.data_as(u64p) for key in keys],
                                 *[x[key].ctypes.data_as(u64p) for key in ("cm_g", "ro", "vo")], C.byref(st))
    if rc not in (0, E_REJECT):
        raise LfPlusError(rc, "lfplus_cm_verify")
    return rc == 0, st.value, x



# ---- ComR1CS / Mlin / PlusProver / PlusVerifier (src/r1cs.rs, lin.rs, mlin.rs, plus.rs) ------------------------------------------------------
@dataclass
class LinParameters:
    """lin.rs:24-28"""
    kappa: int
    decomp: DecompParameters


@dataclass
class PlusParameters:
    """plus.rs:43-47"""
    lin: LinParameters
    B: int


def _centre(x):
    x = np.asarray(x, dtype=np.uint64)
    neg = x > np.uint64(P // 2)
    return np.where(neg, -((np.uint64(P) - x) * neg).astype(np.int64), (x * ~neg).astype(np.int64))


def gadget_decompose(z, b, k):
    """Vec<R>::gadget_decompose(b, k): element j -> its k balanced base-b digits (coefficient-wise, least significant first) at positions [j k, (j + 1) k)"""
    cur = _centre(np.asarray(z, dtype=np.uint64).reshape(-1, D))
    out = np.zeros((cur.shape[0], k, D), dtype=np.int64)
    half = b // 2
    for i in range(k):
        q, rem = np.divmod(cur, b)                       # floor division: rem in [0, b)
        hi = rem > half
        rem, q = np.where(hi, rem - b, rem), np.where(hi, q + 1, q)
        if b % 2 == 0:                                    # |rem| == b / 2 keeps the sign of the value (truncating division in the reference)
            flip = (rem == half) & (cur < 0)
            rem, q = np.where(flip, rem - b, rem), np.where(flip, q + 1, q)
        out[:, i], cur = rem, q
    out = out.reshape(-1, D)
    return np.where(out < 0, np.uint64(P) - np.abs(out).astype(np.uint64), out.astype(np.uint64))


def identity_csr(m):
    """SparseMatrix::identity(m) as (rowptr, col, val[nnz][16])"""
    val = np.zeros((m, D), dtype=np.uint64)
    val[:, 0] = 1
    return np.arange(m + 1, dtype=np.uint32), np.arange(m, dtype=np.uint32), val


def gadget_decompose_csr(mat, b, k):
    """SparseMatrix::gadget_decompose(b, k): rows x m -> rows x (m k) with (M G) gadget_decompose(z) = M z: coefficient c at column j becomes
    c b^i at columns j k + i"""
    rowptr, col, val = (np.asarray(x) for x in mat)
    pw = [pow(b, i, P) for i in range(k)]
    v = np.asarray(val, dtype=np.uint64).astype(object)
    nv = np.stack([(v * pw[i]) % P for i in range(k)], axis=1).astype(np.uint64).reshape(-1, D)
    ncol = (np.asarray(col, dtype=np.uint32)[:, None] * np.uint32(k) + np.arange(k, dtype=np.uint32)[None, :]).reshape(-1)
    return (np.asarray(rowptr, dtype=np.uint32) * np.uint32(k)).astype(np.uint32), ncol.astype(np.uint32), nv


def pad_rows(mat, n):
    rowptr, col, val = mat
    rowptr = np.asarray(rowptr, dtype=np.uint32)
    return np.concatenate([rowptr, np.full(n + 1 - rowptr.size, rowptr[-1], dtype=np.uint32)]), col, val


def r1cs_decomposed_square(r1cs, n, b, k):
    """r1cs.rs:170-184: the three matrices gadget-decomposed (m -> m k columns) and padded to n rows"""
    return tuple(pad_rows(gadget_decompose_csr(m, b, k), n) for m in r1cs)


@dataclass
class ComR1CS:
    """r1cs.rs:21-58: r1cs = (A, B, C) CSR matrices (n x n), z the short witness, f = z.gadget_decompose(b, k), cm_f = A f"""
    r1cs: tuple
    z: np.ndarray
    f: np.ndarray
    cm_f: np.ndarray
    l_in: int = 1

    @staticmethod
    def new(ctx, r1cs, z, l_in, b, k, A=None):
        if A is not None:
            ctx.set_matrix(A)
        f = gadget_decompose(z, b, k)
        return ComR1CS(tuple(r1cs), np.asarray(z, dtype=np.uint64), f, ctx.commit(f), l_in)

    def matrices(self):
        return list(self.r1cs)

    def linearize(self, ctx, transcript, resident=False, preloaded=False, from_f_hint=None):
        """Linearize::linearize (r1cs.rs:76-139) on `ctx` (the witness becomes resident there) -> (LinB fields, ComR1CSProof fields); resident: the three
        matrices are the ones ctx.set_matrices / share_matrices left on the device; preloaded: ctx.set_witness(self.f) was already called (PlusProver.preload);
        from_f_hint: DecompParameters of the Mlin::mlin that follows -- the double commitment of this witness is enqueued on the context's second stream now
        (lfplus_rg_from_f_async) and runs next to the sumcheck rounds"""
        if not preloaded:
            ctx.set_witness(self.f)
        if from_f_hint is not None:
            ctx.rg_from_f_async(from_f_hint)
        n = self.f.shape[0]
        nvars = n.bit_length() - 1
        keep, rp, cp, vp = _csr_args(RESIDENT(3) if resident else self.r1cs)
        msgs, ro, ev = np.zeros((nvars, 4, D), dtype=np.uint64), np.zeros(nvars, dtype=np.uint64), np.zeros((4, D), dtype=np.uint64)
        ctx._chk(_lib().lfplus_r1cs_linearize(ctx.h, transcript.h, rp, cp, vp, *[x.ctypes.data_as(u64p) for x in (msgs, ro, ev)]))
        proof = {"msgs": msgs, "nvars": nvars, "r": ro, "evals": ev}
        linb = {"f": self.f, "cm_f": self.cm_f, "r": ro, "v": ev}
        return linb, proof


def r1cs_verify(transcript, proof, nvars=None):
    """ComR1CSProof::verify (r1cs.rs:141-162), host -> (accepted, stage, ro).  nvars: the VERIFIER's log2 n (None: the proof's own field, as the reference reads
    `self.nvars` -- the arrays are checked against it either way)"""
    nvars = _nvars_ok(proof["nvars"] if nvars is None else nvars)
    msgs, ev = _shaped(proof, "msgs", (nvars, 4, D)), _shaped(proof, "evals", (4, D))
    ro, st = np.zeros(nvars, dtype=np.uint64), C.c_int()
    rc = _lib().lfplus_r1cs_verify(transcript.h, nvars, msgs.ctypes.data_as(u64p), ev.ctypes.data_as(u64p), ro.ctypes.data_as(u64p), C.byref(st))
    if rc not in (0, E_REJECT):
        raise LfPlusError(rc, "lfplus_r1cs_verify")
    return rc == 0, st.value, ro


def decomp_verify(dproof, cm_f, v, B, kappa=None, nM=None):
    """DecompProof::verify (decomp.rs:101-123), host -> (accepted, stage).  kappa / nM: the verifier's (None: the shape of cm_f / v); all six arrays must agree"""
    cm_f, v = np.ascontiguousarray(cm_f, dtype=np.uint64), np.ascontiguousarray(v, dtype=np.uint64)
    kappa = int(cm_f.shape[0] if kappa is None and cm_f.ndim == 2 else (kappa or 0))
    count = int(v.shape[0] if nM is None and v.ndim == 3 else 1 + (nM or 0))
    if kappa < 1 or cm_f.shape != (kappa, D) or v.shape != (count, 2, D):
        raise LfPlusError(E_ARG, "decomp_verify: cm_f must be (kappa, 16) and v (1 + nM, 2, 16)")
    arr = [_shaped(dproof, "C0", (kappa, D)), _shaped(dproof, "C1", (kappa, D)), _shaped(dproof, "v0", (count, 2, D)), _shaped(dproof, "v1", (count, 2, D)), cm_f, v]
    st = C.c_int()
    rc = _lib().lfplus_decomp_verify(arr[0].ctypes.data_as(u64p), arr[1].ctypes.data_as(u64p), arr[0].shape[0], arr[2].ctypes.data_as(u64p),
                                     arr[3].ctypes.data_as(u64p), arr[2].shape[0], arr[4].ctypes.data_as(u64p), arr[5].ctypes.data_as(u64p), B, C.byref(st))
    if rc not in (0, E_REJECT):
        raise LfPlusError(rc, "lfplus_decomp_verify")
    return rc == 0, st.value


def mlin(ctxs, transcript, params, M=()):
    """Mlin{lins, params}.mlin(&A, &M, transcript) (mlin.rs:42-107) over the resident witnesses of ctxs -> (LinB2X fields, CmProof fields).  The folded
    witness g stays on the device as ctxs[0]'s resident witness."""
    L, nM, c0 = len(ctxs), len(M), ctxs[0]
    n, k, kappa, dp = c0.n, params.decomp.k, c0.kappa, params.decomp
    nvars = n.bit_length() - 1
    per = 4 + 4 * nM
    keep, rp, cp, vp = _csr_args(M)
    hs = (C.c_void_p * L)(*[c.h for c in ctxs])
    z = lambda *shape: np.zeros(shape, dtype=np.uint64)
    o = {"r": z(nvars), "msgs": z(nvars, 4, D), "e": z(1 + nM, L * k, D, D), "b": z(L, D), "v": z(L, D), "a": z(L, 1 + nM), "bb": z(L, 1 + nM, D),
         "c": z(L, 1 + nM, D), "comh": z(L, kappa, D), "pa": z(nvars, 3, D), "pb": z(nvars, 3, D), "ea": z(L, per, D), "eb": z(L, per, D),
         "cm_g": z(L, kappa, D), "ro": z(2, nvars), "vo": z(L, 1 + nM, 2, D), "fcoms": z(L, 3, kappa, D)}
    x = {"cm_g": z(kappa, D), "vo": z(1 + nM, 2, D)}
    keys = ("r", "msgs", "e", "b", "v", "a", "bb", "c", "comh", "pa", "pb", "ea", "eb", "cm_g", "ro", "vo", "fcoms")
    c0._chk(_lib().lfplus_mlin(hs, L, transcript.h, dp.b, dp.k, dp.l, nM, rp, cp, vp, *[o[key].ctypes.data_as(u64p) for key in keys],
                               x["cm_g"].ctypes.data_as(u64p), x["vo"].ctypes.data_as(u64p)))
    for c in ctxs:
        c._k = k
    o.update(k=k, ell=dp.l, kappa=kappa, nvars=nvars)
    x["ro"] = o["ro"]
    return x, o


def _ro_pairs(ro):
    """ComX.ro: Vec<(R, R)> -- the two sumcheck points as pairs of ring constants, (nvars, 2, 16)"""
    out = np.zeros((ro.shape[1], 2, D), dtype=np.uint64)
    out[:, 0, 0], out[:, 1, 0] = ro[0], ro[1]
    return out


class PlusProver:
    """plus.rs:15-108.  One context per instance (2 accumulated + ncomp fresh); all share the Ajtai matrix of the first."""

    def __init__(self, A, M, ncomp, params, transcript, device=0, shard=None):
        """shard = (rank, world, transport): a prover column-sharded over `world` ranks, one GPU each -- A is then the rank's column slice (kappa, n / world,
        16); transport is an allgather callable (host transport: PlusContext.set_sharding) or the bytes of an ncclUniqueId (RCCL: PlusContext.dist_init).
        Every rank makes the same calls with the same (whole) witnesses and gets the same proof."""
        self.M, self.params, self.transcript = list(M), params, transcript
        self.ctxs = [PlusContext(device) for _ in range(2 + ncomp)]
        if shard is not None:
            rank, world, transport = shard
            if transport == "model":
                self.ctxs[0].set_sharding_model(rank, world)
            elif callable(transport):
                self.ctxs[0].set_sharding(rank, world, transport)
            else:
                self.ctxs[0].dist_init(rank, world, transport)
        self.ctxs[0].set_matrix(A)
        self.ctxs[0].set_matrices(self.M)
        for c in self.ctxs[1:]:
            c.share_matrix(self.ctxs[0])
            c.share_matrices(self.ctxs[0])
        self.res = RESIDENT(len(self.M))
        self.acc = []          # the accumulated LinB witnesses: host copies of F0, F1 -- or, with device_acc, the markers "ctx0" / "ctx1" (they live in ctxs[0] / ctxs[1])
        self.device_acc = False   # True: the accumulator never leaves the device between proves (lfplus_decompose_resident); accumulator() reads it back
        self._preloaded = None
        self.failed = None     # set when a prove() raised half way: the Fiat-Shamir transcript has advanced and the contexts hold a half-folded state

    @staticmethod
    def init(A, M, ncomp, params, transcript, device=0, shard=None):
        return PlusProver(A, M, ncomp, params, transcript, device, shard)

    def close(self):
        for c in reversed(self.ctxs):
            c.close()
        self.ctxs = []

    def preload(self, comp):
        """upload the witnesses of the instances the next prove(comp) will fold into their contexts now (a pipeline has them on the device already: the timed region
        of bench.py starts after this, as the contract's "inputs resident in HBM" asks)"""
        nacc = len(self.acc)
        if nacc + len(comp) > len(self.ctxs):
            raise LfPlusError(E_ARG, "PlusProver.preload: more instances than contexts (ncomp)")
        for i, ci in enumerate(comp):
            self.ctxs[nacc + i].set_witness(ci.f)
        self._preloaded = [ci.f for ci in comp]

    def accumulator(self):
        """(F0, F1) of the last prove as host arrays"""
        if self.device_acc and self.acc:
            return self.ctxs[0].get_witness(), self.ctxs[1].get_witness()
        return tuple(self.acc)

    def prove(self, comp):
        """PlusProver::prove (plus.rs:77-108) -> PlusProof fields: linb2x, lproof, cmproof, dproof"""
        if self.failed is not None:
            raise LfPlusError(E_ARG, f"PlusProver.prove: this prover failed in an earlier prove() ({self.failed}); its transcript and accumulator are half-advanced -- "
                                     "build a new prover")
        nacc = len(self.acc)
        if nacc + len(comp) > len(self.ctxs):
            raise LfPlusError(E_ARG, "PlusProver.prove: more instances than contexts (ncomp)")
        try:
            return self._prove(comp, nacc)
        except Exception as ex:      # the reference's prove() panics on these paths; here the object survives, so it must refuse to continue from this state
            self.failed = repr(ex)
            raise

    def _prove(self, comp, nacc):
        ctxs = self.ctxs[:nacc + len(comp)]
        lproof = []
        pre = self._preloaded is not None and len(self._preloaded) == len(comp) and all(a is ci.f for a, ci in zip(self._preloaded, comp))
        self._preloaded = None
        # RgInstance::from_f of every instance (Mlin::mlin, mlin.rs:52-60) needs no challenge: each is enqueued on its context's second stream as soon as the
        # witness is resident and runs next to the linearizations' latency-bound rounds; lfplus_mlin collects the results
        dp = self.params.lin.decomp
        # Witnesses that are still on the host (the accumulator of a prover without device_acc, fresh instances that were not preloaded) cross PCIe on a worker
        # thread, one after the other, while this thread linearizes the instances that have arrived (an upload is 2.4 ms per 2^20-row witness, a linearization
        # 2.3 ms): only the first upload is exposed.  Sharded provers upload in line (their contexts exchange from this thread).
        ups = [(ctxs[i], f) for i, f in enumerate(self.acc) if not isinstance(f, str)]      # (device_acc: F0 / F1 of the last prove are resident in ctxs[0] / ctxs[1] already)
        if not pre:
            ups += [(ctxs[nacc + i], ci.f) for i, ci in enumerate(comp)]
        arrived, up_err, outs = {}, [], []
        want_outs = not self.device_acc      # (F0, F1) come back to the host: their buffers are allocated AND touched by the worker while the GPU works
        if (ups or want_outs) and self.ctxs[0].shard[1] == 1 and not os.environ.get("LFPLUS_SERIAL_UPLOADS"):
            import threading
            for cx, _ in ups:
                arrived[id(cx)] = threading.Event()

            def _upload():
                try:
                    for cx, f in ups:
                        cx.set_witness(f)
                        arrived[id(cx)].set()
                    if want_outs:
                        outs.extend(np.zeros((ctxs[0].n, D), dtype=np.uint64) for _ in range(2))
                        for b in outs:
                            b[::32, 0] = 0           # one word per 4 KiB page (32 rows of 128 bytes): the pages exist before the download needs them
                        outs_ready.set()
                except Exception as ex:      # (reported by the thread that waits for the witness)
                    up_err.append(ex)
                    for ev in arrived.values():
                        ev.set()
            outs_ready = threading.Event()
            th = threading.Thread(target=_upload, daemon=True)
            th.start()
        else:
            th = None
            for cx, f in ups:
                cx.set_witness(f)

        def _wait(cx):
            ev = arrived.get(id(cx))
            if ev is not None:
                ev.wait()
                if up_err:
                    raise up_err[0]
        try:
            for i in range(nacc):
                _wait(ctxs[i])
                ctxs[i].rg_from_f_async(dp)
            for i, ci in enumerate(comp):
                same = len(self.M) == 3 and all(a is b for x, y in zip(ci.r1cs, self.M) for a, b in zip(x, y))   # (M = cr1cs.x.matrices() in every reference use)
                _wait(ctxs[nacc + i])
                _, lp = ci.linearize(ctxs[nacc + i], self.transcript, resident=same, preloaded=True, from_f_hint=dp)
                lproof.append(lp)
        except Exception:
            if th is not None:               # (never leave the worker writing into contexts the caller is about to close)
                th.join()
            raise
        linb2x, cmproof = mlin(ctxs, self.transcript, self.params.lin, self.res)
        if self.device_acc:
            dec = ctxs[0].decompose(None, None, self.params.B, _ro_pairs(linb2x["ro"]), self.res, into=(self.ctxs[0], self.ctxs[1]))
            self.acc = ["ctx0", "ctx1"]
        else:
            if th is not None:
                th.join()
                if up_err:
                    raise up_err[0]
            dec = ctxs[0].decompose(None, None, self.params.B, _ro_pairs(linb2x["ro"]), self.res, out_bufs=tuple(outs) if len(outs) == 2 else None)
            self.acc = [dec["F0"], dec["F1"]]
        dproof = {key: dec[key] for key in ("C0", "C1", "v0", "v1")}
        return {"linb2x": linb2x, "lproof": lproof, "cmproof": cmproof, "dproof": dproof}


class PlusVerifier:
    """plus.rs:25-33, 110-146 (host only).  verify returns True, or False with .stage = (which proof, stage) -- the reference panics there"""

    def __init__(self, A, M, params, transcript):
        self.M, self.params, self.transcript = list(M), params, transcript
        self.stage = None
        # the verifier's OWN shape parameters (plus.rs:110-146 passes &self.A, &self.M down; CmProof::verify sizes everything from them, cm.rs:349-365):
        # nothing below is taken from the proof
        A = np.asarray(A)
        if A.ndim != 3 or A.shape[2] != D or A.shape[0] != params.lin.kappa or A.shape[1] < 2 or A.shape[1] & (A.shape[1] - 1):
            raise LfPlusError(E_ARG, "PlusVerifier: A must be (kappa, n = 2^nvars, 16) with kappa = params.lin.kappa")
        self.kappa, self.n = int(A.shape[0]), int(A.shape[1])
        self.nvars = self.n.bit_length() - 1

    @staticmethod
    def init(A, M, params, transcript):
        return PlusVerifier(A, M, params, transcript)

    def verify(self, proof):
        """False with .stage = (which proof, stage); a proof whose arrays do not have the shapes this verifier's parameters imply is rejected as
        ("malformed", message) before any C verifier reads it"""
        dp, nM = self.params.lin.decomp, len(self.M)
        try:
            for i, lp in enumerate(proof["lproof"]):
                ok, st, _ = r1cs_verify(self.transcript, lp, nvars=self.nvars)
                if not ok:
                    self.stage = (f"lproof[{i}]", st)
                    return False
            cm = proof["cmproof"]
            L = int(np.asarray(cm["b"]).shape[0])      # the number of folded instances is the statement's (the reference: self.lins.len()); every array must agree with it
            fcoms = np.ascontiguousarray(cm["fcoms"], dtype=np.uint64)
            if fcoms.shape != (L, 3, self.kappa, D):
                raise LfPlusError(E_ARG, "malformed proof: fcoms")
            ok, st, _ = cm_verify(self.transcript, cm, list(fcoms), nvars=self.nvars, L=L, k=dp.k, ell=dp.l, kappa=self.kappa, nM=nM)
            if not ok:
                self.stage = ("cmproof", st)
                return False
            ok, st = decomp_verify(proof["dproof"], proof["linb2x"]["cm_g"], proof["linb2x"]["vo"], self.params.B, kappa=self.kappa, nM=nM)
            if not ok:
                self.stage = ("dproof", st)
                return False
        except LfPlusError as ex:
            if ex.code != E_ARG:
                raise
            self.stage = ("malformed", str(ex))
            return False
        except (KeyError, TypeError, IndexError) as ex:
            self.stage = ("malformed", repr(ex))
            return False
        self.stage = None
        return True


# ---- deterministic synthetic workloads (numpy only: no oracle, no GPU) -----------------------------------------------------------------------
# The shape of the reference's end-to-end bench (benches/e2e.rs:57-100: `setup_input(n, L, k, kappa)`: decomposed-square R1CS over identity matrices,
# BinaryChoice witness, L fresh instances folded by one PlusProver::prove, B = estimate_bound(d * 128, L, d, k) / 2, l = ceil(log_{d/2} q)).
# Differences, deliberate: the L instances are distinct (the bench clones one), the Ajtai matrix is i.i.d. from an indexable SplitMix64 stream.
PLUS_CONFIGS = {
    # name: (nvars, L, k, kappa)
    "P12": (12, 2, 2, 1),        # tiny: unit tests of the sharded path (kappa k d l d = 11264 > n: from_f refuses it -- used with k = 1 shapes only)
    "P15": (15, 3, 4, 1),        # kappa 1 keeps tau inside n = 2^15 (kappa k d l d = 22528)
    "P16": (16, 3, 4, 2),        # benches/utils/mod.rs:282-301 PROTOCOL_SCALING[1] = (65536, 3, 4, 2)
    "P17": (17, 3, 4, 2),        # PROTOCOL_SCALING[2] = (131072, 3, 4, 2): the reference's largest e2e row
    "P20": (20, 3, 4, 2),        # BASELINE configs[4]: 2^20 rows (the same row extrapolated)
}


def estimate_bound(sop, L, d, k):
    """utils::estimate_bound (utils.rs:102-112)"""
    a, c = sop * L, d // 2 + d * k + 1
    return math.ceil((a + math.sqrt(float(a * a + 4 * a * c))) / 2.0)


def _splitmix_words(seed, start, count):
    from .workload import _G, _M1, _M2
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & (2**64 - 1)) + idx * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


@dataclass
class PlusWorkload:
    name: str
    nvars: int
    L: int
    k: int
    kappa: int
    B: int
    l: int

    @property
    def n(self):
        return 1 << self.nvars

    def params(self):
        return PlusParameters(LinParameters(self.kappa, DecompParameters(D // 2, self.k, self.l)), self.B)

    def ajtai_matrix(self, cols=None):
        """(kappa, n, 16) canonical words, or the column range cols = (c0, c1) of it (what a rank of a sharded prover uploads)"""
        c0, c1 = cols if cols is not None else (0, self.n)
        rows = [_splitmix_words(0xA17A2 + self.nvars, (i * self.n + c0) * D, (c1 - c0) * D) % np.uint64(P) for i in range(self.kappa)]
        return np.stack(rows).reshape(self.kappa, c1 - c0, D)

    def r1cs(self):
        return r1cs_decomposed_square((identity_csr(self.n // self.k),) * 3, self.n, self.B, self.k)

    def z(self, i):
        """WitnessPattern::BinaryChoice: constant coefficient 0 / 1 (so z o z = z holds for the square system)"""
        z = np.zeros((self.n // self.k, D), dtype=np.uint64)
        z[:, 0] = _splitmix_words(0x2B1A0 + 977 * i + self.nvars, 0, self.n // self.k) >> np.uint64(63)
        return z


def make_plus_workload(name):
    nvars, L, k, kappa = PLUS_CONFIGS[name]
    B = estimate_bound(D * 128, L, D, k) // 2                                    # benches/e2e.rs:71
    return PlusWorkload(name, nvars, L, k, kappa, B, math.ceil(math.log(float(P)) / math.log(D / 2)))
