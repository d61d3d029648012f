"""ctypes binding of the LatticeFold+ oracle slice (oracle/liblfp.so: FrogRing RqPoly, coefficient form).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "..", "oracle")
_SO = os.path.join(_DIR, "liblfp.so")
P = 15912092521325583641
D = 16
u64p = C.POINTER(C.c_uint64)
_lib = None


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_DIR, f) for f in ("lfp.c", "lfp.h")]
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
            subprocess.check_call(["make", "-C", _DIR, "-s", "liblfp.so"])
        L = C.CDLL(_SO)
        L.lfp_ring_mul.argtypes = [u64p, u64p, u64p]
        L.lfp_tensor_product.argtypes = [u64p, C.c_size_t, u64p, C.c_size_t, u64p]
        L.lfp_tensor.argtypes = [u64p, C.c_size_t, u64p]
        L.lfp_exp.argtypes = [C.c_int64, u64p]
        L.lfp_commit.argtypes = [u64p, C.c_uint32, C.c_size_t, u64p, u64p]
        L.lfp_rg_from_f.argtypes = [u64p, C.c_size_t, u64p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_int8), u64p, u64p, u64p, u64p, u64p]
        L.lfp_splitmix_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, u64p]
        u32pp, u64pp = C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64p)
        L.lfp_decompose.argtypes = [u64p, C.c_size_t, u64p, C.c_uint32, C.c_uint64, u64p, u64p, C.c_uint32, u32pp, u32pp, u64pp, u64p, u64p, u64p, u64p, u64p, u64p]
        for f in ("lfp_ring_mul", "lfp_tensor_product", "lfp_tensor", "lfp_commit", "lfp_splitmix_fill"):
            getattr(L, f).restype = None
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def fq(xs):
    return np.array([int(x) % P for x in xs], dtype=np.uint64)


def ring_mul(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    o = np.zeros(D, dtype=np.uint64)
    lib().lfp_ring_mul(_p(a), _p(b), _p(o))
    return o


def tensor_product(a, b):
    a, b = fq(a), fq(b)
    o = np.zeros(max(1, a.size * b.size if a.size and b.size else a.size + b.size), dtype=np.uint64)
    lib().lfp_tensor_product(_p(a), a.size, _p(b), b.size, _p(o))
    return o


def tensor(r):
    r = fq(r)
    o = np.zeros(1 << r.size, dtype=np.uint64)
    lib().lfp_tensor(_p(r), r.size, _p(o))
    return o


def splitmix(seed, start, count):
    o = np.empty(count, dtype=np.uint64)
    lib().lfp_splitmix_fill(seed, start, count, _p(o))
    return o


def commit(A, f):
    A = np.ascontiguousarray(A, dtype=np.uint64)
    f = np.ascontiguousarray(f, dtype=np.uint64)
    kappa, n = A.shape[0], A.shape[1]
    o = np.zeros((kappa, D), dtype=np.uint64)
    lib().lfp_commit(_p(A.reshape(-1)), kappa, n, _p(f.reshape(-1)), _p(o.reshape(-1)))
    return o


def rg_from_f(f, A, b, k, l):
    """RgInstance::from_f -> dict(Df [k][n][16] int8, comMf [k][kappa][16][16], tau [n], cm_f, C_Mf, cm_mtau [kappa][16])"""
    A = np.ascontiguousarray(A, dtype=np.uint64)
    f = np.ascontiguousarray(f, dtype=np.uint64)
    kappa, n = A.shape[0], A.shape[1]
    Df = np.zeros((k, n, D), dtype=np.int8)
    com = np.zeros((k, kappa, D, D), dtype=np.uint64)
    tau = np.zeros(n, dtype=np.uint64)
    cmf, cmM, cmt = (np.zeros((kappa, D), dtype=np.uint64) for _ in range(3))
    rc = lib().lfp_rg_from_f(_p(f.reshape(-1)), n, _p(A.reshape(-1)), kappa, b, k, l, Df.ctypes.data_as(C.POINTER(C.c_int8)), _p(com.reshape(-1)),
                             _p(tau), _p(cmf.reshape(-1)), _p(cmM.reshape(-1)), _p(cmt.reshape(-1)))
    if rc != 0:
        raise ValueError(f"lfp_rg_from_f: {rc}")
    return {"Df": Df, "comMf": com, "tau": tau, "cm_f": cmf, "C_Mf": cmM, "cm_mtau": cmt}


def csr_args(mats):
    """mats: list of (rowptr uint32 [n+1], col uint32 [nnz], val uint64 [nnz][16]) -> ctypes pointer arrays (and the arrays, to keep them alive)"""
    keep = [(np.ascontiguousarray(r, dtype=np.uint32), np.ascontiguousarray(c, dtype=np.uint32), np.ascontiguousarray(v, dtype=np.uint64)) for r, c, v in mats]
    u32p = C.POINTER(C.c_uint32)
    rp = (u32p * max(1, len(keep)))(*[k[0].ctypes.data_as(u32p) for k in keep])
    cp = (u32p * max(1, len(keep)))(*[k[1].ctypes.data_as(u32p) for k in keep])
    vp = (u64p * max(1, len(keep)))(*[k[2].ctypes.data_as(u64p) for k in keep])
    return keep, rp, cp, vp


def decompose(f, A, B, r_a, r_b, mats=()):
    """Decomp::decompose -> dict(F0, F1 (n,16); C0, C1 (kappa,16); v0, v1 (1+nm, 2, 16))"""
    f = np.ascontiguousarray(f, dtype=np.uint64)
    A = np.ascontiguousarray(A, dtype=np.uint64)
    r_a = np.ascontiguousarray(r_a, dtype=np.uint64)
    r_b = np.ascontiguousarray(r_b, dtype=np.uint64)
    kappa, n = A.shape[0], A.shape[1]
    nm = len(mats)
    keep, rp, cp, vp = csr_args(mats)
    F0, F1 = np.zeros((n, D), dtype=np.uint64), np.zeros((n, D), dtype=np.uint64)
    C0, C1 = np.zeros((kappa, D), dtype=np.uint64), np.zeros((kappa, D), dtype=np.uint64)
    v0, v1 = np.zeros((1 + nm, 2, D), dtype=np.uint64), np.zeros((1 + nm, 2, D), dtype=np.uint64)
    rc = lib().lfp_decompose(_p(f.reshape(-1)), n, _p(A.reshape(-1)), kappa, B, _p(r_a.reshape(-1)), _p(r_b.reshape(-1)), nm, rp, cp, vp,
                             _p(F0.reshape(-1)), _p(F1.reshape(-1)), _p(C0.reshape(-1)), _p(C1.reshape(-1)), _p(v0.reshape(-1)), _p(v1.reshape(-1)))
    if rc != 0:
        raise ValueError(f"lfp_decompose: {rc}")
    return {"F0": F0, "F1": F1, "C0": C0, "C1": C1, "v0": v0, "v1": v1}
