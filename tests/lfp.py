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
        srcs = [os.path.join(_DIR, f) for f in ("lfp.c", "lfp_protocol.c", "lfp.h")]
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


# ---- transcript-driven part (oracle/lfp_protocol.c): PoseidonTranscript<RqPoly>, set check, range check ------------------------------
def _plib():
    L = lib()
    if not getattr(L, "_proto", False):
        vp = C.c_void_p
        u8p = C.POINTER(C.c_uint8)
        u32pp, u64pp = C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64p)
        L.lfp_tr_new.restype = vp
        L.lfp_tr_clone.restype = vp
        L.lfp_tr_clone.argtypes = [vp]
        L.lfp_tr_free.argtypes = [vp]
        L.lfp_tr_absorb.argtypes = [vp, u64p, C.c_size_t]
        L.lfp_tr_challenge.argtypes = [vp]
        L.lfp_tr_challenge.restype = C.c_uint64
        L.lfp_tr_squeeze_bytes.argtypes = [vp, C.c_size_t, u8p]
        L.lfp_short_challenge.argtypes = [vp, u64p]
        L.lfp_short_challenge_from_bytes.argtypes = [u8p, u64p]
        L.lfp_poseidon_params.argtypes = [u64p, u64p]
        L.lfp_poseidon_permute.argtypes = [u64p]
        L.lfp_psi.argtypes = [u64p]
        L.lfp_ct_psi_mul.argtypes = [u64p]
        L.lfp_ct_psi_mul.restype = C.c_uint64
        L.lfp_set_check.argtypes = [vp, C.c_uint, u64p, C.c_uint, C.c_uint, u64p, C.c_uint, C.c_uint, u32pp, u32pp, u64pp, u64p, u64p, u64p, u64p]
        L.lfp_set_check_verify.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, u64p, u64p, u64p, u64p]
        L.lfp_range_check.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, u64pp, u64pp, u64pp, u64pp, C.c_uint, u32pp, u32pp, u64pp] + [u64p] * 8
        L.lfp_range_check_verify.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint] + [u64p] * 8
        L._proto = True
    return L


class Transcript:
    """PoseidonTranscript::empty::<FrogPoseidonConfig>() (latticefold-plus/src/transcript.rs)"""

    def __init__(self, h=None):
        self.h = h if h is not None else _plib().lfp_tr_new()

    def clone(self):
        return Transcript(_plib().lfp_tr_clone(self.h))

    def absorb(self, ring):
        a = np.ascontiguousarray(ring, dtype=np.uint64).reshape(-1, D)
        _plib().lfp_tr_absorb(self.h, _p(a.reshape(-1)), a.shape[0])

    def challenge(self):
        return int(_plib().lfp_tr_challenge(self.h))

    def squeeze_bytes(self, n):
        o = np.zeros(n, dtype=np.uint8)
        _plib().lfp_tr_squeeze_bytes(self.h, n, o.ctypes.data_as(C.POINTER(C.c_uint8)))
        return o

    def short_challenge(self):
        o = np.zeros(D, dtype=np.uint64)
        _plib().lfp_short_challenge(self.h, _p(o))
        return o

    def __del__(self):
        try:
            _plib().lfp_tr_free(self.h)
        except Exception:
            pass


def poseidon_params():
    ark, mds = np.zeros(720, dtype=np.uint64), np.zeros(576, dtype=np.uint64)
    _plib().lfp_poseidon_params(_p(ark), _p(mds))
    return ark, mds


def short_challenge_from_bytes(bs):
    b = np.ascontiguousarray(bs, dtype=np.uint8)
    o = np.zeros(D, dtype=np.uint64)
    _plib().lfp_short_challenge_from_bytes(b.ctypes.data_as(C.POINTER(C.c_uint8)), _p(o))
    return o


def psi():
    o = np.zeros(D, dtype=np.uint64)
    _plib().lfp_psi(_p(o))
    return o


def exp_dense(digits):
    """exp() of an int8 array of digits in (-8, 8): one-hot ring elements, shape digits.shape + (16,)"""
    d = np.asarray(digits).astype(np.int64)
    e = np.where(d >= 0, d, D + d)
    out = np.zeros(d.shape + (D,), dtype=np.uint64)
    np.put_along_axis(out, e[..., None], 1, axis=-1)
    return out


def set_check(tr, nvars, msets, vsets=None, mats=()):
    """In::set_check.  msets (nmat, n, ncols, 16), vsets (nvec, n, 16) -> dict(r, msgs (nvars,4,16), e (1+nM, nmat, ncols, 16), b (nvec,16))"""
    msets = np.ascontiguousarray(msets, dtype=np.uint64)
    nmat, n, ncols = msets.shape[:3]
    vsets = np.zeros((0, n, D), dtype=np.uint64) if vsets is None else np.ascontiguousarray(vsets, dtype=np.uint64)
    nvec, nM = vsets.shape[0], len(mats)
    keep, rp, cp, vp = csr_args(mats)
    r, msgs = np.zeros(nvars, dtype=np.uint64), np.zeros((nvars, 4, D), dtype=np.uint64)
    e, b = np.zeros((1 + nM, nmat, ncols, D), dtype=np.uint64), np.zeros((max(nvec, 1), D), dtype=np.uint64)
    vs = vsets if nvec else np.zeros((1, 1, D), dtype=np.uint64)
    rc = _plib().lfp_set_check(tr.h, nvars, _p(msets.reshape(-1)), nmat, ncols, _p(vs.reshape(-1)), nvec, nM, rp, cp, vp, _p(r), _p(msgs.reshape(-1)),
                               _p(e.reshape(-1)), _p(b.reshape(-1)))
    if rc:
        raise ValueError(f"lfp_set_check: {rc}")
    return {"r": r, "msgs": msgs, "e": e, "b": b[:nvec]}


def set_check_verify(tr, nvars, out, nM=0):
    e, b, msgs = (np.ascontiguousarray(out[k], dtype=np.uint64) for k in ("e", "b", "msgs"))
    nmat, ncols, nvec = e.shape[1], e.shape[2], b.shape[0]
    bb = b if nvec else np.zeros((1, D), dtype=np.uint64)
    r = np.zeros(nvars, dtype=np.uint64)
    return _plib().lfp_set_check_verify(tr.h, nvars, nmat, ncols, nvec, nM, _p(msgs.reshape(-1)), _p(e.reshape(-1)), _p(bb.reshape(-1)), _p(r)), r


def range_check(tr, nvars, instances, k, mats=()):
    """Rg::range_check.  instances: dicts with Mf (k, n, 16, 16), tau (n,), mtau (n, 16), f (n, 16) -> dict of the Dcom fields"""
    L, nM = len(instances), len(mats)
    keep, rp, cp, vp = csr_args(mats)
    arrs = {key: [np.ascontiguousarray(i[key], dtype=np.uint64) for i in instances] for key in ("Mf", "tau", "mtau", "f")}
    ptrs = {key: (u64p * L)(*[_p(a.reshape(-1)) for a in v]) for key, v in arrs.items()}
    r, msgs = np.zeros(nvars, dtype=np.uint64), np.zeros((nvars, 4, D), dtype=np.uint64)
    e, b = np.zeros((1 + nM, L * k, D, D), dtype=np.uint64), np.zeros((L, D), dtype=np.uint64)
    v, a = np.zeros((L, D), dtype=np.uint64), np.zeros((L, 1 + nM), dtype=np.uint64)
    bb, c = np.zeros((L, 1 + nM, D), dtype=np.uint64), np.zeros((L, 1 + nM, D), dtype=np.uint64)
    rc = _plib().lfp_range_check(tr.h, nvars, L, k, ptrs["Mf"], ptrs["tau"], ptrs["mtau"], ptrs["f"], nM, rp, cp, vp, _p(r), _p(msgs.reshape(-1)),
                                 _p(e.reshape(-1)), _p(b.reshape(-1)), _p(v.reshape(-1)), _p(a.reshape(-1)), _p(bb.reshape(-1)), _p(c.reshape(-1)))
    if rc:
        raise ValueError(f"lfp_range_check: {rc}")
    return {"r": r, "msgs": msgs, "e": e, "b": b, "v": v, "a": a, "bb": bb, "c": c}


def range_check_verify(tr, nvars, d, k):
    arr = {key: np.ascontiguousarray(d[key], dtype=np.uint64) for key in ("msgs", "e", "b", "v", "a", "bb", "c")}
    L, nM = arr["b"].shape[0], arr["a"].shape[1] - 1
    r = np.zeros(nvars, dtype=np.uint64)
    return _plib().lfp_range_check_verify(tr.h, nvars, L, k, nM, *[_p(arr[key].reshape(-1)) for key in ("msgs", "e", "b", "v", "a", "bb", "c")], _p(r)), r


def cm_prove(tr, nvars, instances, k, ell, kappa, mats=()):
    """Cm::prove (cm.rs:56-347).  instances: dicts with Mf, tau, mtau, f (as range_check) + comMf (k, kappa, 16, 16), fcoms (3, kappa, 16: cm_f | C_Mf | cm_mtau)"""
    L, nM = len(instances), len(mats)
    n = 1 << nvars
    keep, rp, cp, vp = csr_args(mats)
    keys = ("Mf", "tau", "mtau", "f", "comMf", "fcoms")
    arrs = {key: [np.ascontiguousarray(i[key], dtype=np.uint64) for i in instances] for key in keys}
    ptrs = {key: (u64p * L)(*[_p(a.reshape(-1)) for a in v]) for key, v in arrs.items()}
    per = 4 + 4 * nM
    o = {"r": np.zeros(nvars, dtype=np.uint64), "msgs": np.zeros((nvars, 4, D), dtype=np.uint64), "e": np.zeros((1 + nM, L * k, D, D), dtype=np.uint64),
         "b": np.zeros((L, D), dtype=np.uint64), "v": np.zeros((L, D), dtype=np.uint64), "a": np.zeros((L, 1 + nM), dtype=np.uint64),
         "bb": np.zeros((L, 1 + nM, D), dtype=np.uint64), "c": np.zeros((L, 1 + nM, D), dtype=np.uint64), "comh": np.zeros((L, kappa, D), dtype=np.uint64),
         "pa": np.zeros((nvars, 3, D), dtype=np.uint64), "pb": np.zeros((nvars, 3, D), dtype=np.uint64), "ea": np.zeros((L, per, D), dtype=np.uint64),
         "eb": np.zeros((L, per, D), dtype=np.uint64), "g": np.zeros((L, n, D), dtype=np.uint64), "cm_g": np.zeros((L, kappa, D), dtype=np.uint64),
         "ro": np.zeros((2, nvars), dtype=np.uint64), "vo": np.zeros((L, 1 + nM, 2, D), dtype=np.uint64)}
    L_ = _plib()
    u32pp, u64pp = C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64p)
    L_.lfp_cm_prove.argtypes = [C.c_void_p] + [C.c_uint] * 5 + [u64pp] * 6 + [C.c_uint, u32pp, u32pp, u64pp] + [u64p] * 17
    rc = L_.lfp_cm_prove(tr.h, nvars, L, k, ell, kappa, *[ptrs[key] for key in keys], nM, rp, cp, vp,
                         *[_p(o[key].reshape(-1)) for key in ("r", "msgs", "e", "b", "v", "a", "bb", "c", "comh", "pa", "pb", "ea", "eb", "g", "cm_g", "ro", "vo")])
    if rc:
        raise ValueError(f"lfp_cm_prove: {rc}")
    o.update(k=k, ell=ell, kappa=kappa, nvars=nvars)
    return o


def cm_verify(tr, proof, fcoms):
    """CmProof::verify (cm.rs:349-580) -> (rc, dict(cm_g, ro, vo))"""
    nvars, k, ell, kappa = proof["nvars"], proof["k"], proof["ell"], proof["kappa"]
    L, nM = proof["b"].shape[0], proof["a"].shape[1] - 1
    fc = [np.ascontiguousarray(x, dtype=np.uint64) for x in fcoms]
    fptr = (u64p * L)(*[_p(x.reshape(-1)) for x in fc])
    x = {"cm_g": np.zeros((L, kappa, D), dtype=np.uint64), "ro": np.zeros((2, nvars), dtype=np.uint64), "vo": np.zeros((L, 1 + nM, 2, D), dtype=np.uint64)}
    L_ = _plib()
    L_.lfp_cm_verify.argtypes = [C.c_void_p] + [C.c_uint] * 6 + [C.POINTER(u64p)] + [u64p] * 15
    arr = {key: np.ascontiguousarray(proof[key], dtype=np.uint64) for key in ("msgs", "e", "b", "v", "a", "bb", "c", "comh", "pa", "pb", "ea", "eb")}
    rc = L_.lfp_cm_verify(tr.h, nvars, L, k, ell, kappa, nM, fptr, *[_p(arr[key].reshape(-1)) for key in ("msgs", "e", "b", "v", "a", "bb", "c", "comh", "pa", "pb", "ea", "eb")],
                          _p(x["cm_g"].reshape(-1)), _p(x["ro"].reshape(-1)), _p(x["vo"].reshape(-1)))
    return rc, x


# ---- ComR1CS::linearize, Mlin::mlin, PlusProver::prove, PlusVerifier::verify (r1cs.rs, mlin.rs, plus.rs) restated over the oracle's C pieces --------
def r1cs_linearize(tr, nvars, f, r1cs):
    keep, rp, cp, vp = csr_args(r1cs)
    f = np.ascontiguousarray(f, dtype=np.uint64)
    o = {"msgs": np.zeros((nvars, 4, D), dtype=np.uint64), "r": np.zeros(nvars, dtype=np.uint64), "evals": np.zeros((4, D), dtype=np.uint64)}
    L_ = _plib()
    u32pp, u64pp = C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64p)
    L_.lfp_r1cs_linearize.argtypes = [C.c_void_p, C.c_uint, u64p, u32pp, u32pp, u64pp, u64p, u64p, u64p]
    rc = L_.lfp_r1cs_linearize(tr.h, nvars, _p(f.reshape(-1)), rp, cp, vp, _p(o["msgs"].reshape(-1)), _p(o["r"]), _p(o["evals"].reshape(-1)))
    assert rc == 0
    o["nvars"] = nvars
    return o


def r1cs_verify(tr, proof):
    L_ = _plib()
    L_.lfp_r1cs_verify.argtypes = [C.c_void_p, C.c_uint, u64p, u64p, u64p]
    msgs, ev = (np.ascontiguousarray(proof[key], dtype=np.uint64) for key in ("msgs", "evals"))
    ro = np.zeros(proof["nvars"], dtype=np.uint64)
    return L_.lfp_r1cs_verify(tr.h, proof["nvars"], _p(msgs.reshape(-1)), _p(ev.reshape(-1)), _p(ro)), ro


def decomp_verify(dproof, cm_f, v, B):
    L_ = _plib()
    L_.lfp_decomp_verify.argtypes = [u64p, u64p, C.c_uint, u64p, u64p, C.c_uint, u64p, u64p, C.c_uint64]
    arr = [np.ascontiguousarray(x, dtype=np.uint64) for x in (dproof["C0"], dproof["C1"], dproof["v0"], dproof["v1"], cm_f, v)]
    return L_.lfp_decomp_verify(_p(arr[0].reshape(-1)), _p(arr[1].reshape(-1)), arr[0].shape[0], _p(arr[2].reshape(-1)), _p(arr[3].reshape(-1)), arr[2].shape[0],
                                _p(arr[4].reshape(-1)), _p(arr[5].reshape(-1)), B)


def gadget_decompose(z, b, k):
    """Vec<R>::gadget_decompose(b, k) through lfp_balanced_digits (element j -> positions [j k, (j + 1) k), least significant digit first)"""
    z = np.asarray(z, dtype=np.uint64).reshape(-1, D)
    L_ = lib()
    L_.lfp_balanced_digits.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.POINTER(C.c_int64)]
    table = {}
    for v in np.unique(z):
        dg = (C.c_int64 * k)()
        L_.lfp_balanced_digits(int(v), b, k, dg)
        table[int(v)] = [int(x) % P for x in dg]
    out = np.zeros((z.shape[0], k, D), dtype=np.uint64)
    for v, dg in table.items():
        mask = z == np.uint64(v)
        for i in range(k):
            out[:, i][mask] = dg[i]
    return out.reshape(-1, D)


def identity_csr(m):
    val = np.zeros((m, D), dtype=np.uint64)
    val[:, 0] = 1
    return np.arange(m + 1, dtype=np.uint32), np.arange(m, dtype=np.uint32), val


def r1cs_decomposed_square(r1cs, n, b, k):
    """r1cs.rs:170-184 with SparseMatrix::gadget_decompose = right-multiplication by the gadget matrix (coefficient c at column j -> c b^i at j k + i)"""
    out = []
    for rowptr, col, val in r1cs:
        nr, nc, nv = [0], [], []
        for r in range(len(rowptr) - 1):
            for t in range(rowptr[r], rowptr[r + 1]):
                for i in range(k):
                    nc.append(int(col[t]) * k + i)
                    nv.append([(int(x) * pow(b, i, P)) % P for x in val[t]])
            nr.append(len(nc))
        nr += [len(nc)] * (n + 1 - len(nr))
        out.append((np.array(nr, dtype=np.uint32), np.array(nc, dtype=np.uint32), np.array(nv, dtype=np.uint64).reshape(-1, D)))
    return tuple(out)


def addmod(a, b):
    s = a + b                                     # wraps mod 2^64; a, b < p < 2^64 < 2p
    return np.where((s < a) | (s >= np.uint64(P)), s - np.uint64(P), s)


class PlusOracle:
    """PlusProver (plus.rs:49-108): linearize every fresh instance, Mlin::mlin over accumulated + fresh, Decomp::decompose of the folded witness"""

    def __init__(self, A, M, kappa, b, k, l, B, tr):
        self.A, self.M, self.kappa, self.b, self.k, self.l, self.B, self.tr = np.ascontiguousarray(A, dtype=np.uint64), list(M), kappa, b, k, l, B, tr
        self.acc = []

    def prove(self, comps):
        n = self.A.shape[1]
        nvars = n.bit_length() - 1
        lproof = [r1cs_linearize(self.tr, nvars, f, r1cs) for f, r1cs in comps]
        lins = self.acc + [f for f, _ in comps]
        insts = []
        for f in lins:
            rg = rg_from_f(f, self.A, self.b, self.k, self.l)
            tau_i = np.array([int(t) if int(t) <= P // 2 else int(t) - P for t in rg["tau"]], dtype=np.int8)
            insts.append({"Mf": exp_dense(rg["Df"]), "tau": rg["tau"], "mtau": exp_dense(tau_i), "f": f, "comMf": rg["comMf"],
                          "fcoms": np.stack([rg["cm_f"], rg["C_Mf"], rg["cm_mtau"]])})
        cm = cm_prove(self.tr, nvars, insts, self.k, self.l, self.kappa, self.M)
        cm["fcoms"] = np.stack([i["fcoms"] for i in insts])
        g, cm_g, vo = cm["g"][0], cm["cm_g"][0], cm["vo"][0]
        for i in range(1, len(lins)):
            g, cm_g, vo = addmod(g, cm["g"][i]), addmod(cm_g, cm["cm_g"][i]), addmod(vo, cm["vo"][i])
        r_a, r_b = np.zeros((nvars, D), dtype=np.uint64), np.zeros((nvars, D), dtype=np.uint64)
        r_a[:, 0], r_b[:, 0] = cm["ro"][0], cm["ro"][1]
        dec = decompose(g, self.A, self.B, r_a, r_b, self.M)
        self.acc = [dec["F0"], dec["F1"]]
        return {"linb2x": {"cm_g": cm_g, "ro": cm["ro"], "vo": vo}, "lproof": lproof, "cmproof": cm, "dproof": {key: dec[key] for key in ("C0", "C1", "v0", "v1")},
                "g": g}


def plus_verify(tr, proof, B):
    """PlusVerifier::verify (plus.rs:134-146) -> 0, or (which, rc)"""
    for i, lp in enumerate(proof["lproof"]):
        rc = r1cs_verify(tr, lp)[0]
        if rc:
            return (f"lproof[{i}]", rc)
    rc = cm_verify(tr, proof["cmproof"], proof["cmproof"]["fcoms"])[0]
    if rc:
        return ("cmproof", rc)
    rc = decomp_verify(proof["dproof"], proof["linb2x"]["cm_g"], proof["linb2x"]["vo"], B)
    return ("dproof", rc) if rc else 0
