"""ctypes binding of the CPU oracle (oracle/liblfo.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
This module binds the GoldilocksRingNTT build; tests/lfo_bb.py re-executes the same source with
_RING = "babybear" to bind oracle/liblfo_bb.so (BabyBearRingNTT, d = 72, tau = 9).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(_HERE, "..", "oracle")
_RING = globals().get("_RING", "goldilocks")
_SO = os.path.join(_ORACLE_DIR, "liblfo.so" if _RING == "goldilocks" else "liblfo_bb.so")

P, RE, TAU = {"goldilocks": (2**64 - 2**32 + 1, 24, 3), "babybear": (15 * 2**27 + 1, 72, 9)}[_RING]
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def build(force=False):
    srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _SO


class Params(C.Structure):
    _fields_ = [("s", C.c_uint32), ("wit_len", C.c_uint32), ("l", C.c_uint32), ("L", C.c_uint32),
                ("K", C.c_uint32), ("b", C.c_uint32), ("B", C.c_uint64), ("kappa", C.c_uint32),
                ("t", C.c_uint32), ("q", C.c_uint32), ("d", C.c_uint32)]


class Ccs(C.Structure):
    _fields_ = [("rowptr", C.POINTER(u32p)), ("col", C.POINTER(u32p)), ("val", C.POINTER(u64p)),
                ("S_off", u32p), ("S_idx", u32p), ("c", u64p)]


def _p64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


_lib = None


def default_threads():
    """OpenMP threads for the oracle: the usable cores (cgroup/affinity aware), capped at 32 -- the restatement's
    parallel loops are short and 256 spinning threads on a quota-limited box make it crawl."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def lib():
    global _lib
    if _lib is None:
        os.environ.setdefault("OMP_NUM_THREADS", str(default_threads()))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        build()
        L = C.CDLL(_SO)
        L.lfo_modulus.restype = C.c_uint64
        assert L.lfo_modulus() == P and L.lfo_ring_degree() == RE and L.lfo_ring_tau() == TAU
        L.lfo_num_threads.restype = C.c_int
        L.lfo_set_num_threads.argtypes = [C.c_int]
        L.lfo_set_num_threads(int(os.environ["OMP_NUM_THREADS"]))  # explicit: another runtime (torch) may have initialised OpenMP already
        L.lfo_set_ring.restype = C.c_int
        L.lfo_set_ring.argtypes = [C.c_uint64, u64p]
        L.lfo_lcccs_len.restype = C.c_size_t
        L.lfo_cccs_len.restype = C.c_size_t
        L.lfo_proof_len.restype = C.c_size_t
        L.lfo_transcript_new.restype = C.c_void_p
        for f in ("lfo_transcript_free", "lfo_transcript_absorb_fq", "lfo_transcript_absorb_ring",
                  "lfo_transcript_get_challenge", "lfo_transcript_get_short_challenge"):
            getattr(L, f).restype = None
        L.lfo_transcript_free.argtypes = [C.c_void_p]
        L.lfo_transcript_absorb_fq.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.lfo_transcript_absorb_ring.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.lfo_transcript_get_challenge.argtypes = [C.c_void_p, u64p]
        L.lfo_transcript_get_short_challenge.argtypes = [C.c_void_p, u64p]
        L.lfo_crt.argtypes = [u64p, u64p, C.c_size_t]
        L.lfo_icrt.argtypes = [u64p, u64p, C.c_size_t]
        L.lfo_decompose.argtypes = [u64p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int, u64p]
        L.lfo_recompose.argtypes = [u64p, C.c_size_t, C.c_uint64, C.c_uint32, u64p]
        L.lfo_ajtai_commit.argtypes = [u64p, C.c_uint32, C.c_size_t, u64p, u64p]
        L.lfo_build_eq.argtypes = [u64p, C.c_uint32, u64p]
        L.lfo_mle_eval.argtypes = [u64p, C.c_size_t, u64p, C.c_uint32, u64p]
        L.lfo_rot_lin_combination.argtypes = [u64p, u64p, C.c_uint32, C.c_uint32, u64p]
        L.lfo_short_challenge_from_bytes.argtypes = [C.c_char_p, C.c_size_t, u64p]
        L.lfo_witness_from_w_ccs.argtypes = [C.POINTER(Params), u64p, u64p]
        L.lfo_linearize.argtypes = [C.POINTER(Params), C.POINTER(Ccs), C.c_void_p, u64p, u64p, u64p, u64p]
        L.lfo_fold_step.argtypes = [C.POINTER(Params), C.POINTER(Ccs), u64p, C.c_void_p, u64p, u64p, u64p, u64p,
                                    u64p, u64p, u64p]
        L.lfo_verify.argtypes = [C.POINTER(Params), C.POINTER(Ccs), C.c_void_p, u64p, u64p, u64p, u64p]
        L.lfo_decomposition_prove.argtypes = [C.POINTER(Params), C.POINTER(Ccs), u64p, C.c_void_p, u64p, u64p, u64p, u64p]
        L.lfo_set_ring_general.restype = C.c_int
        L.lfo_set_ring_general.argtypes = [u64p, u64p]
        L.lfo_get_ring.restype = None
        L.lfo_get_ring.argtypes = [u64p, u64p]
        L.lfo_set_digit_mode.argtypes = [C.c_int]
        L.lfo_set_digit_mode.restype = None
        L.lfo_sumcheck_fold.argtypes = [C.POINTER(Params), C.c_void_p, u64p, u64p, u64p, u64p]
        L.lfo_horner_combine.argtypes = [u64p, C.c_uint32, C.c_uint32, C.c_size_t, u64p, u64p]
        L.lfo_horner_combine.restype = None
        L.lfo_lincomb.argtypes = [u64p, u64p, C.c_uint32, C.c_size_t, u64p]
        L.lfo_lincomb.restype = None
        L.lfo_splitmix_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, u64p]
        L.lfo_splitmix_fill.restype = None
        _lib = L
    return _lib


# ---------------------------------------------------------------------------------------------
class Transcript:
    def __init__(self):
        self.h = lib().lfo_transcript_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().lfo_transcript_free(self.h)
            self.h = None

    def absorb_fq(self, xs):
        a = np.ascontiguousarray(xs, dtype=np.uint64)
        lib().lfo_transcript_absorb_fq(self.h, _p64(a), a.size)

    def absorb_ring(self, e):
        a = np.ascontiguousarray(e, dtype=np.uint64).reshape(-1)
        assert a.size % RE == 0
        lib().lfo_transcript_absorb_ring(self.h, _p64(a), a.size // RE)

    def challenge(self):
        o = np.zeros(TAU, dtype=np.uint64)
        lib().lfo_transcript_get_challenge(self.h, _p64(o))
        return o

    def short_challenge(self):
        o = np.zeros(RE, dtype=np.uint64)
        lib().lfo_transcript_get_short_challenge(self.h, _p64(o))
        return o


def crt(x):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    o = np.empty_like(x)
    lib().lfo_crt(_p64(x), _p64(o), x.size // RE)
    return o


def icrt(x):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    o = np.empty_like(x)
    lib().lfo_icrt(_p64(x), _p64(o), x.size // RE)
    return o


def decompose(x, base, digits, layout):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    cnt = x.size // RE
    o = np.zeros(cnt * digits * RE, dtype=np.uint64)
    lib().lfo_decompose(_p64(x), cnt, base, digits, layout, _p64(o))
    return o.reshape(-1, RE)


def recompose(x, base, digits):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    cnt = x.size // RE // digits
    o = np.zeros(cnt * RE, dtype=np.uint64)
    lib().lfo_recompose(_p64(x), cnt, base, digits, _p64(o))
    return o.reshape(-1, RE)


def ajtai_commit(A, kappa, n, f):
    A = np.ascontiguousarray(A, dtype=np.uint64)
    f = np.ascontiguousarray(f, dtype=np.uint64)
    o = np.zeros(kappa * RE, dtype=np.uint64)
    lib().lfo_ajtai_commit(_p64(A), kappa, n, _p64(f), _p64(o))
    return o.reshape(kappa, RE)


def build_eq(r_ring):
    r = np.ascontiguousarray(r_ring, dtype=np.uint64).reshape(-1, RE)
    nv = r.shape[0]
    o = np.zeros((1 << nv) * RE, dtype=np.uint64)
    lib().lfo_build_eq(_p64(r), nv, _p64(o))
    return o.reshape(-1, RE)


def mle_eval(table, r_ring):
    t = np.ascontiguousarray(table, dtype=np.uint64).reshape(-1, RE)
    r = np.ascontiguousarray(r_ring, dtype=np.uint64).reshape(-1, RE)
    o = np.zeros(RE, dtype=np.uint64)
    lib().lfo_mle_eval(_p64(t), t.shape[0], _p64(r), r.shape[0], _p64(o))
    return o


def rot_lin_combination(rho_coeff, theta, n, tau_elems=TAU):
    a = np.ascontiguousarray(rho_coeff, dtype=np.uint64).reshape(-1)
    b = np.ascontiguousarray(theta, dtype=np.uint64).reshape(-1)
    o = np.zeros(tau_elems * RE, dtype=np.uint64)
    lib().lfo_rot_lin_combination(_p64(a), _p64(b), n, tau_elems, _p64(o))
    return o.reshape(tau_elems, RE)


def short_challenge_from_bytes(bs):
    o = np.zeros(RE, dtype=np.uint64)
    rc = lib().lfo_short_challenge_from_bytes(bytes(bs), len(bs), _p64(o))
    assert rc == 0
    return o


# ---------------------------------------------------------------------------------------------
class Instance:
    """Holds Params + CCS (CSR) + Ajtai matrix in the oracle's C layout."""

    def __init__(self, wl):
        """wl: latticefold_amd.workload.Workload"""
        self.wl = wl
        self.params = Params(wl.s, wl.wit_len, wl.l, wl.L, wl.K, wl.b, wl.B, wl.kappa, wl.t, wl.q, wl.d)
        t = wl.t
        self._rp = [np.ascontiguousarray(a, dtype=np.uint32) for a in wl.rowptr]
        self._ci = [np.ascontiguousarray(a, dtype=np.uint32) for a in wl.col]
        self._va = [np.ascontiguousarray(a, dtype=np.uint64).reshape(-1) for a in wl.val]
        self._soff = np.ascontiguousarray(wl.S_off, dtype=np.uint32)
        self._sidx = np.ascontiguousarray(wl.S_idx, dtype=np.uint32)
        self._c = np.ascontiguousarray(wl.c, dtype=np.uint64).reshape(-1)
        self._rpp = (u32p * t)(*[_p32(a) for a in self._rp])
        self._cip = (u32p * t)(*[_p32(a) for a in self._ci])
        self._vap = (u64p * t)(*[_p64(a) for a in self._va])
        self.ccs = Ccs(C.cast(self._rpp, C.POINTER(u32p)), C.cast(self._cip, C.POINTER(u32p)),
                       C.cast(self._vap, C.POINTER(u64p)), _p32(self._soff), _p32(self._sidx), _p64(self._c))
        self.lcccs_len = lib().lfo_lcccs_len(C.byref(self.params))
        self.cccs_len = lib().lfo_cccs_len(C.byref(self.params))
        self.proof_len = lib().lfo_proof_len(C.byref(self.params))

    def witness_from_w_ccs(self, w_ccs):
        w = np.ascontiguousarray(w_ccs, dtype=np.uint64).reshape(-1)
        o = np.zeros(self.wl.N * RE, dtype=np.uint64)
        lib().lfo_witness_from_w_ccs(C.byref(self.params), _p64(w), _p64(o))
        return o.reshape(-1, RE)

    def linearize(self, tr, cccs, f_coeff):
        cccs = np.ascontiguousarray(cccs, dtype=np.uint64).reshape(-1)
        f = np.ascontiguousarray(f_coeff, dtype=np.uint64).reshape(-1)
        lc = np.zeros(self.lcccs_len * RE, dtype=np.uint64)
        wl = self.wl
        pr = np.zeros((wl.s * (wl.d + 2) + TAU + wl.t) * RE, dtype=np.uint64)
        rc = lib().lfo_linearize(C.byref(self.params), C.byref(self.ccs), tr.h, _p64(cccs), _p64(f), _p64(lc), _p64(pr))
        assert rc == 0, rc
        return lc.reshape(-1, RE), pr.reshape(-1, RE)

    def fold_step(self, tr, A, acc, w_acc, cm_i, w_i):
        A = np.ascontiguousarray(A, dtype=np.uint64).reshape(-1)
        acc = np.ascontiguousarray(acc, dtype=np.uint64).reshape(-1)
        w_acc = np.ascontiguousarray(w_acc, dtype=np.uint64).reshape(-1)
        cm_i = np.ascontiguousarray(cm_i, dtype=np.uint64).reshape(-1)
        w_i = np.ascontiguousarray(w_i, dtype=np.uint64).reshape(-1)
        lc = np.zeros(self.lcccs_len * RE, dtype=np.uint64)
        f0 = np.zeros(self.wl.N * RE, dtype=np.uint64)
        pr = np.zeros(self.proof_len * RE, dtype=np.uint64)
        rc = lib().lfo_fold_step(C.byref(self.params), C.byref(self.ccs), _p64(A), tr.h, _p64(acc), _p64(w_acc),
                                 _p64(cm_i), _p64(w_i), _p64(lc), _p64(f0), _p64(pr))
        assert rc == 0, rc
        return lc.reshape(-1, RE), f0.reshape(-1, RE), pr.reshape(-1, RE)

    def decomposition_prove(self, tr, A, lcccs, f_coeff):
        """LFDecompositionProver::prove alone -> (K decomposed LCCCS flat, decomposition proof flat)"""
        A = np.ascontiguousarray(A, dtype=np.uint64).reshape(-1)
        lcccs = np.ascontiguousarray(lcccs, dtype=np.uint64).reshape(-1)
        f = np.ascontiguousarray(f_coeff, dtype=np.uint64).reshape(-1)
        wl = self.wl
        pr = np.zeros(wl.K * (wl.t + TAU + wl.l + 1 + wl.kappa) * RE, dtype=np.uint64)
        lcs = np.zeros(wl.K * self.lcccs_len * RE, dtype=np.uint64)
        rc = lib().lfo_decomposition_prove(C.byref(self.params), C.byref(self.ccs), _p64(A), tr.h, _p64(lcccs), _p64(f), _p64(pr), _p64(lcs))
        assert rc == 0, rc
        return lcs.reshape(-1, RE), pr.reshape(-1, RE)

    def sumcheck_fold(self, tr, tables, mu_ring):
        """the folding sumcheck on a caller-supplied mle list -> (msgs [s*(2b+1)], point [s]) as ring elements"""
        t = np.ascontiguousarray(tables, dtype=np.uint64).reshape(-1)
        mu = np.ascontiguousarray(mu_ring, dtype=np.uint64).reshape(-1)
        wl = self.wl
        msgs = np.zeros(wl.s * (2 * wl.b + 1) * RE, dtype=np.uint64)
        pt = np.zeros(wl.s * RE, dtype=np.uint64)
        rc = lib().lfo_sumcheck_fold(C.byref(self.params), tr.h, _p64(t), _p64(mu), _p64(msgs), _p64(pt))
        assert rc == 0
        return msgs.reshape(-1, RE), pt.reshape(-1, RE)

    def ajtai_matrix(self):
        """the workload's synthetic Ajtai matrix (workload.Workload.ajtai_matrix) generated by the oracle's C splitmix"""
        wl = self.wl
        out = np.empty(wl.kappa * wl.N * RE, dtype=np.uint64)
        lib().lfo_splitmix_fill(wl.ajtai_seed(), 0, out.size, _p64(out))
        return out.reshape(wl.kappa, wl.N, RE)

    def verify(self, tr, acc, cm_i, proof):
        acc = np.ascontiguousarray(acc, dtype=np.uint64).reshape(-1)
        cm_i = np.ascontiguousarray(cm_i, dtype=np.uint64).reshape(-1)
        proof = np.ascontiguousarray(proof, dtype=np.uint64).reshape(-1)
        lc = np.zeros(self.lcccs_len * RE, dtype=np.uint64)
        rc = lib().lfo_verify(C.byref(self.params), C.byref(self.ccs), tr.h, _p64(acc), _p64(cm_i), _p64(proof), _p64(lc))
        return rc, lc.reshape(-1, RE)


def horner_combine(tables, ch_ring):
    t = np.ascontiguousarray(tables, dtype=np.uint64)
    g, pg, ln = t.shape[0], t.shape[1], t.shape[2]
    ch = np.ascontiguousarray(ch_ring, dtype=np.uint64).reshape(-1)
    o = np.zeros(ln * RE, dtype=np.uint64)
    lib().lfo_horner_combine(_p64(t.reshape(-1)), g, pg, ln, _p64(ch), _p64(o))
    return o.reshape(-1, RE)


def lincomb(coef, tables):
    t = np.ascontiguousarray(tables, dtype=np.uint64)
    n, ln = t.shape[0], t.shape[1]
    cf = np.ascontiguousarray(coef, dtype=np.uint64).reshape(-1)
    o = np.zeros(ln * RE, dtype=np.uint64)
    lib().lfo_lincomb(_p64(cf), _p64(t.reshape(-1)), n, ln, _p64(o))
    return o.reshape(-1, RE)


def get_ring():
    """(nonres, y[8][TAU]) of the binomial form currently installed"""
    nr = C.c_uint64()
    y = np.zeros(8 * TAU, dtype=np.uint64)
    lib().lfo_get_ring(C.cast(C.byref(nr), u64p), _p64(y))
    return int(nr.value), y.reshape(8, TAU)


def set_ring(nonres, y):
    a = np.ascontiguousarray(y, dtype=np.uint64).reshape(-1)
    return lib().lfo_set_ring(int(nonres), _p64(a))


def set_ring_general(crt_matrix, tensor):
    """dense CRT matrix (RE x RE) + structure tensor (TAU^3) -- SURVEY 8(c)'s data form; set_ring(..) goes back"""
    a = np.ascontiguousarray(crt_matrix, dtype=np.uint64).reshape(-1)
    t = np.ascontiguousarray(tensor, dtype=np.uint64).reshape(-1)
    assert a.size == RE * RE and t.size == TAU ** 3
    return lib().lfo_set_ring_general(_p64(a), _p64(t))


def set_digit_mode(mode):
    lib().lfo_set_digit_mode(int(mode))
