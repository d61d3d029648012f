"""Deterministic synthetic LatticeFold workloads (numpy only; no oracle, no GPU).

Mirrors the shape of the reference's own benchmark generator
(crates/latticefold/benches/utils.rs:136-171 "non-scalar" R1CS, arith/r1cs.rs:155-201,289-308):
x_len = 1, x = (1), h = 1, R1CS with A = B = identity rows (1 nnz/row), C = diag(z), padded to
m = N = wit_len*L rows; S = {{0,1},{2}}, c = (1,-1).  Differences, all deliberate:
the witness slots and the Ajtai matrix are i.i.d. uniform from an indexable SplitMix64 stream
(the reference's `AjtaiCommitmentScheme::rand` yields ONE element repeated kappa*n times,
commitment_scheme.rs:30-32).  The same stream is produced on-device by lf_util_fill_uniform.
"""
from dataclasses import dataclass, field

import numpy as np

P = 2**64 - 2**32 + 1
RE = 24  # u64 words per ring element (8 slots x 3 coords / 24 coefficients)

# ring id -> (modulus, ring degree d = words per element, tau); ids match include/lfhip.h LF_RING_*
RINGS = {"goldilocks": (P, 24, 3), "babybear": (15 * 2**27 + 1, 72, 9)}
RING_IDS = {"goldilocks": 0, "babybear": 1}

CONFIGS = {
    # name: (s, wit_len, L, B, b, K, kappa)   -- SURVEY.md 8.0
    "T8": (8, 64, 4, 1 << 16, 2, 16, 4),       # tiny, unit tests
    "T10": (10, 256, 4, 1 << 16, 2, 16, 6),    # small parity case
    "C1": (10, 256, 4, 1 << 16, 2, 16, 21),    # BASELINE configs[0]
    "T12": (12, 1024, 4, 1 << 16, 2, 16, 8),
    "T14": (14, 4096, 4, 1 << 16, 2, 16, 12),
    "C2": (16, 1 << 14, 4, 1 << 16, 2, 16, 25),  # BASELINE configs[1]
    "T18": (18, 1 << 16, 4, 1 << 16, 2, 16, 26),
    "C4": (20, 1 << 18, 4, 1 << 16, 2, 16, 26),  # BASELINE configs[3] / metric config
    # the reference's own wider Goldilocks rows at small wit_len (benches/config.toml:150-165): kappa 42-44 / B 2^22 / L 3 / K 22,
    # kappa 99 (row-chunked commits), and the widest digits the int32 witness planes hold (B 2^31, K 31)
    "E22": (10, 256, 3, 1 << 22, 2, 22, 43),
    "E99": (9, 64, 4, 1 << 16, 2, 16, 99),
    "E31": (9, 64, 3, 1 << 31, 2, 31, 50),
    "E32": (9, 64, 2, 1 << 32, 2, 32, 99),      # benches/config.toml:158: kappa 99, B 2^32, K 32
    # GoldilocksDP of the reference unit tests (decomposition_parameters.rs:89-96): N not a power of 2
    "G5": (9, 64, 5, 1 << 15, 2, 15, 5),
    # n = l + 1 + wit_len = 5120 / 10240 columns: step counts where the chunking of the int8 inner products is not monotone (80 steps ->
    # 40 chunks, 81 -> 27; lf_dot_i8.hip) -- the scratch must be sized for the larger count
    "D5120": (15, 5118, 4, 1 << 16, 2, 16, 5),
    "D10240": (15, 10238, 3, 1 << 22, 2, 22, 5),
    # ---- BabyBearRingNTT (d = 72, tau = 9; B^L = 2^32 > p)
    "B6": (6, 32, 2, 1 << 16, 2, 16, 3, "babybear"),      # tiny, unit tests
    "B8": (8, 128, 2, 1 << 16, 2, 16, 4, "babybear"),
    "B10": (10, 512, 2, 1 << 16, 2, 16, 6, "babybear"),
    "B14": (14, 1 << 13, 2, 1 << 16, 2, 16, 16, "babybear"),
    "C3": (18, 1 << 17, 2, 1 << 16, 2, 16, 16, "babybear"),  # BASELINE configs[2]
    # BabyBearDP of the reference unit tests (decomposition_parameters.rs:98-105): B 2^8, L 4, K 8
    "BDP": (7, 32, 4, 1 << 8, 2, 8, 4, "babybear"),
    "B21": (8, 128, 2, 1 << 16, 2, 16, 21, "babybear"),    # kappa > 16: two row chunks of the int8 commit kernel (11 + 10 rows -> 3 row tiles each)
    "B32": (7, 64, 2, 1 << 16, 2, 16, 32, "babybear"),     # the backend's largest kappa: 2 x 16 rows
    # shapes of the specialised BabyBear commit kernel (k_ajtai_i8x: 4 row tiles = kappa 13..16): a padded row tile (kappa 13), one plane group only (K 8 -> 7
    # planes), a ragged last column tile (N = 2 * 333 columns)
    "B13": (8, 128, 2, 1 << 16, 2, 16, 13, "babybear"),
    "BK8": (9, 128, 4, 1 << 8, 2, 8, 16, "babybear"),
    "B333": (10, 333, 2, 1 << 16, 2, 16, 15, "babybear"),
    "BD768": (12, 766, 2, 1 << 16, 2, 16, 4, "babybear"),   # n = 768: 12 chunks of the int8 inner products where n + 1 gives 7 (bb_dot_i8.hip)
}

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def splitmix_fq(seed: int, start: int, count: int, ring: str = "goldilocks") -> np.ndarray:
    """count canonical residues; word i = splitmix64(seed + (start+i+1)*G) folded into [0,p) (Goldilocks: one
    conditional subtraction; BabyBear: the top 32 bits of the word mod p)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & (2**64 - 1)) + idx * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
        if ring == "goldilocks":
            z = np.where(z >= np.uint64(P), z - np.uint64(P), z)
        else:
            z = (z >> np.uint64(32)) % np.uint64(RINGS[ring][0])
    return z


def diag(v: int, ring: str = "goldilocks") -> np.ndarray:
    """R::from(u128): every slot = (v,0,..,0)."""
    p, d, tau = RINGS[ring]
    e = np.zeros(d, dtype=np.uint64)
    e[0::tau] = np.uint64(v % p)
    return e


def default_nonres(ring: str) -> int:
    """extension-field non-residue of the DEFAULT ring tables (DESIGN.md "CRT map is data"): 2^40 for Goldilocks
    (F_{p^3} = F_p[Y]/(Y^3 - 2^40)), 2 for BabyBear (F_{p^9} = F_p[Y]/(Y^9 - 2))"""
    return 1 << 40 if ring == "goldilocks" else 2


_EPS = np.uint64(0xFFFFFFFF)
_M32 = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def _gl_mul_block(a, b):
    with np.errstate(over="ignore"):
        a0, a1, b0, b1 = a & _M32, a >> _S32, b & _M32, b >> _S32
        lo, m1, m2, hi = a0 * b0, a1 * b0, a0 * b1, a1 * b1
        mid = m1 + m2
        cmid = (mid < m1).astype(np.uint64)
        L = lo + (mid << _S32)
        c1 = (L < lo).astype(np.uint64)
        H = hi + (mid >> _S32) + (cmid << _S32) + c1                   # the 128-bit product is (H, L); H < 2^64 always
        h0, h1 = H & _M32, H >> _S32
        t0 = L - h1
        t0 = np.where(L < h1, t0 - _EPS, t0)                            # borrow: 2^64 = 2^32 - 1
        t1 = h0 * _EPS
        r = t0 + t1
        r = np.where(r < t1, r + _EPS, r)
        r = np.where(r >= np.uint64(P), r - np.uint64(P), r)
    return r


def gl_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """element-wise product of canonical Goldilocks residues (uint64 arrays, broadcasting), vectorised: 32-bit limbs, then 2^64 = 2^32 - 1, 2^96 = -1 (mod p).
    Large operands go through in cache-sized blocks of the leading axis (the ~25 temporaries of a 10^8-element call thrash memory: 130 s against 6 s)."""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    shape = np.broadcast_shapes(a.shape, b.shape)
    if len(shape) == 0 or int(np.prod(shape)) <= (1 << 20):
        return _gl_mul_block(a, b)
    a = np.broadcast_to(a, shape)
    b = np.broadcast_to(b, shape)
    out = np.empty(shape, dtype=np.uint64)
    per = max(1, int(np.prod(shape[1:])))
    step = max(1, (1 << 19) // per)
    for i in range(0, shape[0], step):
        out[i:i + step] = _gl_mul_block(a[i:i + step], b[i:i + step])
    return out


def _fp_mul(a, b, ring):
    if ring == "goldilocks":
        return gl_mul(a, b)
    p = np.uint64(RINGS[ring][0])
    return (np.asarray(a, dtype=np.uint64) * np.asarray(b, dtype=np.uint64)) % p          # 31-bit residues: the product fits 62 bits


def _fp_add(a, b, ring):
    p = np.uint64(RINGS[ring][0])
    with np.errstate(over="ignore"):
        r = a + b
        if ring == "goldilocks":
            r = np.where((r < a) | (r >= p), r - p, r)
        else:
            r = np.where(r >= p, r - p, r)
    return r


def ring_mul_ntt(a: np.ndarray, b: np.ndarray, ring: str = "goldilocks") -> np.ndarray:
    """slot-wise product of NTT-form ring elements (a, b: (..., d) uint64, broadcasting over the leading axes) in F_{p^tau} = F_p[Y]/(Y^tau - nonres);
    vectorised numpy (schoolbook over the tau coordinates, 32-bit limb products for the 64-bit prime): any size"""
    p, d, tau = RINGS[ring]
    nu = np.uint64(default_nonres(ring))
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    shape = np.broadcast_shapes(a.shape, b.shape)
    a2 = np.broadcast_to(a, shape).reshape(-1, 8, tau)
    b2 = np.broadcast_to(b, shape).reshape(-1, 8, tau)
    out = np.zeros(a2.shape, dtype=np.uint64)
    for i in range(tau):
        for j in range(tau):
            t = _fp_mul(a2[:, :, i], b2[:, :, j], ring)
            if i + j >= tau:
                t = _fp_mul(t, nu, ring)
            k = (i + j) % tau
            out[:, :, k] = _fp_add(out[:, :, k], t, ring)
    return out.reshape(shape)


@dataclass
class Workload:
    name: str
    s: int
    wit_len: int
    L: int
    B: int
    b: int
    K: int
    kappa: int
    l: int = 1
    t: int = 3
    q: int = 2
    d: int = 2
    seed: int = 0
    ring: str = "goldilocks"
    rowptr: list = field(default_factory=list)
    col: list = field(default_factory=list)
    val: list = field(default_factory=list)
    S_off: np.ndarray = None
    S_idx: np.ndarray = None
    c: np.ndarray = None
    x_ccs: np.ndarray = None
    w_ccs: np.ndarray = None

    @property
    def N(self):
        return self.wit_len * self.L

    @property
    def m(self):
        return 1 << self.s

    @property
    def n(self):
        return self.l + 1 + self.wit_len

    @property
    def tau(self):
        return RINGS[self.ring][2]

    @property
    def RE(self):
        return RINGS[self.ring][1]

    @property
    def P(self):
        return RINGS[self.ring][0]

    @property
    def ring_id(self):
        return RING_IDS[self.ring]

    def ajtai_seed(self):
        return 0xA17A1 + self.seed

    def ajtai_matrix(self, row0=0, rows=None) -> np.ndarray:
        """kappa x N ring elements (NTT form), i.i.d. uniform words."""
        rows = self.kappa - row0 if rows is None else rows
        per_row = self.N * self.RE
        return splitmix_fq(self.ajtai_seed(), row0 * per_row, rows * per_row, self.ring).reshape(rows, self.N, self.RE)

    def z(self) -> np.ndarray:
        return np.concatenate([self.x_ccs, diag(1, self.ring)[None, :], self.w_ccs], axis=0)

    def alg_bytes(self) -> int:
        """ALGORITHMIC bytes of one fold step, SURVEY.md 8(d) formula (E = 192 B Goldilocks, 288 B BabyBear as u32)."""
        E = 192 if self.ring == "goldilocks" else 288
        N, t, K, L, kap, tau = self.m, self.t, self.K, self.L, self.kappa, self.tau
        P_L, P_F = t + 1, 5 + 2 * K * tau
        lin = t * N * E + (N // L) * E + 3 * P_L * N * E + (tau + t) * N * E
        dec = ((1 + K) * N * E + 2 * K * N * E + (kap + K - 1) * N * E + K * N * E
               + (K * t + K) * N * E + K * t * N * E)
        fold = ((2 * K * t + 2) * N * E + (2 * K + 2) * N * E + 3 * P_F * N * E
                + (2 * K + 2 * K * t) * N * E + (2 * K + 1) * N * E + 2 * N * E)
        return lin + 2 * dec + fold


def chain_w_ccs(wl, j: int) -> np.ndarray:
    """The witness of step j >= 1 of an IVC-style chain under the workload's FIXED constraint system (bench.py --chain, tests/test_gpu_chain.py,
    tests/tools/make_chain_digests.py).  The bench R1CS is A = B = I, C = diag(z_base) (arith/r1cs.rs:170-186): row i demands z_i * z_i = z_base_i * z_i
    slot by slot, so every vector whose slots are each either 0 or the base witness's slot satisfies it.  Step j keeps slot k of element i iff bit k of
    splitmix64(0x1C0C + 1000 j + seed, i) is set (about half of the slots); x_ccs and the constant 1 stay as they are.  j = 0 is the base witness."""
    if j == 0:
        return wl.w_ccs
    _p, RE, tau = RINGS[wl.ring]
    slots = RE // tau
    with np.errstate(over="ignore"):
        idx = np.arange(1, wl.wit_len + 1, dtype=np.uint64)
        z = np.uint64((0x1C0C + 1000 * j + wl.seed) & (2**64 - 1)) + idx * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    keep = ((z[:, None] >> np.arange(slots, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)     # [wit_len][slots]
    return np.where(np.repeat(keep, tau, axis=1), wl.w_ccs, np.uint64(0))


def make_workload(name: str, seed: int = 0, kappa: int = None, ccs: str = "r1cs", l: int = 1) -> Workload:
    """ccs = "r1cs": the reference bench shape (A = B = I, C = diag(z); 1 nnz/row);
       ccs = "deg3": the reference's degree-three non-scalar CCS (arith/ccs.rs:14-43): t = 4, M = (I, I, I, diag(z^2)),
                     S = {{0,1,2},{3}}, c = (1,-1), d = 3;
       ccs = "multi": an R1CS with 2 nnz/row and non-identity values: (A z)_i = z_i + z_{i+1}, B = I,
                      (C z)_i = z_i * (z_i + z_{i+1})  (satisfied by construction);
       ccs = "multi4" / "multi16": 4 / 16 entries per row at pseudo-random columns with scalar coefficients 1..7 in A, ring-valued entries in C (any size:
                      vectorised generators)."""
    cfg = CONFIGS[name]
    s, wit_len, L, B, b, K, kap = cfg[:7]
    ring = cfg[7] if len(cfg) > 7 else "goldilocks"
    p, RE, _tau = RINGS[ring]
    wl = Workload(name=name, s=s, wit_len=wit_len, L=L, B=B, b=b, K=K, kappa=kappa or kap, l=l, seed=seed, ring=ring)
    assert wl.N <= wl.m, "sanity_check (nifs.rs:165-173): m must be >= wit_len*L"
    wl.x_ccs = np.tile(diag(1, ring), (wl.l, 1))
    if wl.l > 1:   # distinct public inputs (slot-constant scalars 2, 3, ..) so that x_s / x_0 handling is really exercised
        for i in range(1, wl.l):
            wl.x_ccs[i] = diag(i + 1, ring)
    wl.w_ccs = splitmix_fq(0x4C460001 + seed, 0, wit_len * RE, ring).reshape(wit_len, RE)
    z = wl.z()
    n, m = wl.n, wl.m
    rows = min(n, m)
    rp = np.minimum(np.arange(m + 1, dtype=np.uint32), np.uint32(rows)).astype(np.uint32)
    ci = np.arange(rows, dtype=np.uint32)
    ident = np.tile(diag(1, ring), (rows, 1))
    wl.rowptr = [rp, rp.copy(), rp.copy()]
    wl.col = [ci, ci.copy(), ci.copy()]
    wl.val = [ident, ident.copy(), np.ascontiguousarray(z[:rows])]
    wl.S_off = np.array([0, 2, 3], dtype=np.uint32)
    wl.S_idx = np.array([0, 1, 2], dtype=np.uint32)
    wl.c = np.stack([diag(1, ring), diag(p - 1, ring)])
    if ccs == "deg3":
        zsq = ring_mul_ntt(z[:rows], z[:rows], ring)
        wl.t, wl.q, wl.d = 4, 2, 3
        wl.rowptr = [rp, rp.copy(), rp.copy(), rp.copy()]
        wl.col = [ci, ci.copy(), ci.copy(), ci.copy()]
        wl.val = [ident, ident.copy(), ident.copy(), np.ascontiguousarray(zsq)]
        wl.S_off = np.array([0, 3, 4], dtype=np.uint32)
        wl.S_idx = np.array([0, 1, 2, 3], dtype=np.uint32)
    elif ccs == "multi":
        assert rows >= 2
        nxt = (np.arange(rows, dtype=np.uint32) + 1) % np.uint32(rows)
        rp2 = np.minimum(2 * np.arange(m + 1, dtype=np.uint64), np.uint64(2 * rows)).astype(np.uint32)
        ci2 = np.stack([ci, nxt], axis=1).reshape(-1).astype(np.uint32)
        vA = np.tile(diag(1, ring), (2 * rows, 1))
        zc = np.repeat(z[:rows], 2, axis=0)                       # C row i: z_i at columns i and i+1
        wl.rowptr = [rp2, rp.copy(), rp2.copy()]
        wl.col = [ci2, ci.copy(), ci2.copy()]
        wl.val = [vA, ident.copy(), np.ascontiguousarray(zc)]
    elif ccs in ("multi4", "multi16"):
        # A general sparse R1CS at any size: k entries per row at pseudo-random columns with small non-unit scalar coefficients,
        #   (A z)_i = sum_j a_ij z_c(i,j),   B = I,   (C z)_i = sum_j (a_ij z_i) z_c(i,j) = z_i (A z)_i      -- satisfied by construction for every witness family
        # that keeps z_i fixed (the values of C are ring elements with eight distinct slots: a genuinely ring-valued SpMV)
        k = int(ccs[5:])
        assert rows >= k
        with np.errstate(over="ignore"):
            idx = np.arange(1, rows * k + 1, dtype=np.uint64)
            h = np.uint64((0xCC5 + seed) & (2**64 - 1)) + idx * _G
            h = (h ^ (h >> np.uint64(30))) * _M1
            h = (h ^ (h >> np.uint64(27))) * _M2
            h = h ^ (h >> np.uint64(31))
        cols = (h % np.uint64(rows)).astype(np.uint32)                      # [rows * k]
        coef = ((h >> np.uint64(40)) % np.uint64(7) + np.uint64(1))          # a_ij in 1 .. 7
        rpk = np.minimum(k * np.arange(m + 1, dtype=np.uint64), np.uint64(k * rows)).astype(np.uint32)
        vA = np.zeros((rows * k, RE), dtype=np.uint64)
        vA[:, 0::_tau] = coef[:, None]
        vC = _fp_mul(np.repeat(z[:rows], k, axis=0), coef[:, None], ring)    # a_ij * z_i, slot by slot (scalar times element)
        wl.rowptr = [rpk, rp.copy(), rpk.copy()]
        wl.col = [cols, ci.copy(), cols.copy()]
        wl.val = [vA, ident.copy(), np.ascontiguousarray(vC)]
    elif ccs != "r1cs":
        raise ValueError(ccs)
    return wl
