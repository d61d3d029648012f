"""R1CS -> CCS front-end (numpy only), mirroring `CCS::from_r1cs_padded`
(crates/latticefold/src/arith.rs:122-172) and `Instance::get_z_vector` (arith.rs:399-409):

    M = (A, B, C),  S = {{0,1},{2}},  c = (1, -1),  t = 3, q = 2, d = 2,
    rows padded to  max((n - l - 1) * L, m).next_power_of_two(),   z = x_ccs || 1 || w_ccs.

Matrices are given as dense integer arrays (small circuits) or (rowptr, col, val) CSR triples with ring-element values.
`vitalik_r1cs` / `vitalik_z` are the reference's own test circuit x^3 + x + 5 = y (arith/r1cs.rs:128-151, 224-262).
"""
import numpy as np

from .workload import RINGS, Workload, diag


def _csr_from_dense(M, ring):
    p, d, tau = RINGS[ring]
    rows, cols = M.shape
    rp, ci, va = [0], [], []
    for r in range(rows):
        for c in range(cols):
            v = int(M[r, c]) % p
            if v:
                ci.append(c)
                va.append(diag(v, ring))
        rp.append(len(ci))
    return (np.array(rp, dtype=np.uint32), np.array(ci, dtype=np.uint32),
            np.array(va, dtype=np.uint64).reshape(len(ci), d))


def _pad_rows(csr, m):
    rp, ci, va = csr
    if len(rp) - 1 < m:
        rp = np.concatenate([rp, np.full(m - (len(rp) - 1), rp[-1], dtype=np.uint32)])
    return rp, ci, va


def workload_from_r1cs(A, B, C, l, x_ccs, w_ccs, *, ring="goldilocks", L, Bbase, b=2, K, kappa, name="r1cs", seed=0, W=None):
    """CCS::from_r1cs_padded(r1cs, W, L) (arith.rs:144-149) + the parameter set, packaged as a Workload (what Context.load_ccs / the
    oracle take).  A, B, C: dense integer matrices (rows x n) or CSR triples; x_ccs: (l, d) ring elements; w_ccs: (wit_len, d).
    W is the reference's caller-supplied row count m of `from_r1cs` (arith.rs:122-140), default = the R1CS row count; the CCS is then
    padded to max((n - l - 1) * L, W).next_power_of_two() rows exactly as the reference does."""
    p, d, tau = RINGS[ring]
    csr = [m if isinstance(m, tuple) else _csr_from_dense(np.asarray(m), ring) for m in (A, B, C)]
    rows = len(csr[0][0]) - 1
    wit_len = len(w_ccs)
    n = l + 1 + wit_len
    W = rows if W is None else int(W)
    assert W >= rows, "W (CCS::from_r1cs's m) must cover the R1CS rows"
    m = max((n - l - 1) * L, W)                 # (ccs.n - ccs.l - 1) * L = wit_len * L
    m = 1 << (m - 1).bit_length()               # next_power_of_two
    s = m.bit_length() - 1
    wl = Workload(name=name, s=s, wit_len=wit_len, L=L, B=Bbase, b=b, K=K, kappa=kappa, l=l, t=3, q=2, d=2, seed=seed, ring=ring)
    csr = [_pad_rows(c, m) for c in csr]
    for rp, ci, va in csr:
        assert ci.size == 0 or int(ci.max()) < n, "column index outside z"
    wl.rowptr = [c[0] for c in csr]
    wl.col = [c[1] for c in csr]
    wl.val = [c[2] for c in csr]
    wl.S_off = np.array([0, 2, 3], dtype=np.uint32)
    wl.S_idx = np.array([0, 1, 2], dtype=np.uint32)
    wl.c = np.stack([diag(1, ring), diag(p - 1, ring)])
    wl.x_ccs = np.ascontiguousarray(x_ccs, dtype=np.uint64).reshape(l, d)
    wl.w_ccs = np.ascontiguousarray(w_ccs, dtype=np.uint64).reshape(wit_len, d)
    return wl


def vitalik_r1cs():
    """R1CS for x^3 + x + 5 = y (the reference's get_test_r1cs, arith/r1cs.rs:128-151); z = (x, 1, y, x^2, x^3, x^3 + x)."""
    A = np.array([[1, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0], [1, 0, 0, 0, 1, 0], [0, 5, 0, 0, 0, 1]])
    B = np.array([[1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0]])
    C = np.array([[0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 1], [0, 0, 1, 0, 0, 0]])
    return A, B, C


def vitalik_z_ntt(ring="goldilocks"):
    """get_test_z_ntt (arith/r1cs.rs:236-262): slot k of every entry carries the scalar assignment for input = k."""
    p, d, tau = RINGS[ring]
    z = np.zeros((6, d), dtype=np.uint64)
    for k in range(8):
        x = k
        vals = [x, 1, x ** 3 + x + 5, x * x, x ** 3, x ** 3 + x]
        for j, v in enumerate(vals):
            z[j, tau * k] = v % p
    return z


def check_r1cs_scalar_slots(A, B, C, z, ring="goldilocks"):
    """(A z) o (B z) == C z slot by slot for assignments whose slots are base-field scalars (host check, Python integers)"""
    p, d, tau = RINGS[ring]
    for k in range(8):
        zs = np.array([int(v) for v in z[:, tau * k]], dtype=object)
        az, bz, cz = (np.asarray(M, dtype=object).dot(zs) % p for M in (A, B, C))
        if any((int(a) * int(b) - int(c)) % p for a, b, c in zip(az, bz, cz)):
            return False
    return True
