"""Calibration of the conventions that no in-tree reference test pins: the CRT slot map, the F_{p^3} non-residue, the basis of the
extension field and the balanced-digit tie rule all live in the absent `stark-rings` crate.  `tools/probe_stark_rings.rs` (run on a machine that has the
reference workspace) prints them as one JSON object; this module reads that object, checks it against the parametrisation the product and the oracle use,
and installs it:

    cal = load_probe("stark_rings_tables.json")        # CalibrationError if the data cannot be expressed through the data APIs
    cal.apply(ctx)                                     # lf_set_ring_tables / lf_set_ext_basis / lf_set_digit_mode on a Goldilocks context
    cal.apply_oracle(lfo)                              # the same conventions on the CPU oracle (tests/lfo.py)

Host-only (numpy and Python integers); nothing here touches a GPU or the oracle library.  The Goldilocks ring only: the probe's BabyBear and Frog entries are
kept verbatim in `Calibration.extra` (their loaders are the same three calls on a BabyBear context / a handful of literals for oracle/lfp.h)."""
import json
import os
from dataclasses import dataclass, field

import numpy as np

P = 18446744069414584321          # Goldilocks
RE, TAU, SLOTS = 24, 3, 8


class CalibrationError(ValueError):
    pass


def _binomial_mul(a, b, nu):
    r = [0] * TAU
    for i in range(TAU):
        for j in range(TAU):
            if i + j < TAU:
                r[i + j] += a[i] * b[j]
            else:
                r[i + j - TAU] += nu * a[i] * b[j]
    return [v % P for v in r]


def balanced_digits(v, base, digits, mode):
    """the balanced base-`base` digits of the canonical residue v, as signed integers (oracle/lfo_ring.c decompose_coeff, lf_set_digit_mode):
    mode 0 = truncate toward zero and move a remainder of magnitude > base/2 to the other side (ties +-base/2 keep the sign of the value);
    mode 1 = floor rule, digits in [-base/2, base/2) (base 2 always takes the mode-0 form: the floor rule does not terminate there)"""
    cur = v if v <= (P - 1) // 2 else v - P
    half = base // 2
    out = []
    for _ in range(digits):
        if mode == 0 or base == 2:
            q = abs(cur) // base * (1 if cur >= 0 else -1)
            rem = cur - q * base
            if abs(rem) > half:
                if rem < 0:
                    rem += base
                    q -= 1
                else:
                    rem -= base
                    q += 1
        else:
            rem = cur % base
            if rem >= half:
                rem -= base
            q = (cur - rem) // base
        out.append(rem)
        cur = q
    return out


@dataclass
class Calibration:
    nonres: int                    # Y^3 of the extension field's generator
    y: np.ndarray                  # (8, 3): the image of X in every CRT slot, internal binomial coordinates
    ext_basis: np.ndarray          # (3, 3) T with external = T internal (identity when the crate's basis is the binomial one)
    digit_mode: int                # 0 / 1 (lf_set_digit_mode)
    serialized_words_le: bool      # the probe's serialized element is its 24 flat words as little-endian u64, nothing else (what lf_wire.cpp assumes)
    digit_mode_ambiguous: bool = False   # the probe's digit cases held no tie / negative-remainder case: both rules reproduce them and mode 0 was taken unverified
    extra: dict = field(default_factory=dict)

    def apply(self, ctx):
        """install on a Goldilocks product context (latticefold_amd.api.Context) -- before any matrix / witness is uploaded"""
        ctx.set_ring_tables(self.nonres, np.ascontiguousarray(self.y, dtype=np.uint64).reshape(-1))
        if not (self.ext_basis == np.eye(TAU, dtype=np.uint64)).all():
            ctx.set_ext_basis(self.ext_basis)
        ctx.set_digit_mode(self.digit_mode)

    def apply_oracle(self, lfo, general_data=None):
        """install on the CPU oracle (tests/lfo.py).  A non-identity basis needs the oracle's general form: pass general_data(nonres, y, T) -> (crt, tensor)
        (tests/test_gpu_ext_basis.py has it) or the data comes out in the internal basis"""
        if (self.ext_basis == np.eye(TAU, dtype=np.uint64)).all() or general_data is None:
            rc = lfo.set_ring(self.nonres, self.y)
        else:
            rc = lfo.set_ring_general(*general_data(self.nonres, self.y, self.ext_basis))
        if rc != 0:
            raise CalibrationError(f"the oracle refused the ring data (rc {rc}): the slot images are not 8 distinct roots of X^24 - X^12 + 1")
        lfo.set_digit_mode(self.digit_mode)


def _ext_basis_from_tensor(tensor, nonres_hint):
    """T from the crate's own multiplication table e_i * e_j (probe entry ext_mul_tensor_goldilocks_fq3): the crate's basis must be a PERMUTED binomial basis
    {g^k} (a tower or a reordering); anything else needs lf_set_ext_basis by hand.  Returns (T, nonres)"""
    t = [[[int(x) % P for x in tensor[i * TAU + j]] for j in range(TAU)] for i in range(TAU)]
    unit = lambda i: [int(k == i) for k in range(TAU)]

    def mul(a, b):
        r = [0] * TAU
        for i in range(TAU):
            for j in range(TAU):
                if a[i] and b[j]:
                    for k in range(TAU):
                        r[k] += a[i] * b[j] * t[i][j][k]
        return [v % P for v in r]
    one = next((i for i in range(TAU) if all(mul(unit(i), unit(j)) == unit(j) for j in range(TAU))), None)
    if one is None:
        raise CalibrationError("ext_mul_tensor: no unit vector is the identity: not a permuted binomial basis (install the basis change by hand: lf_set_ext_basis)")
    for g in range(TAU):
        if g == one:
            continue
        pw, idx, ok = unit(one), [], True
        for _ in range(TAU):
            if sorted(pw) != [0] * (TAU - 1) + [1]:
                ok = False
                break
            idx.append(pw.index(1))
            pw = mul(pw, unit(g))
        if not ok or len(set(idx)) != TAU:
            continue
        if any(pw[k] for k in range(TAU) if k != one):      # g^tau must be in the base field
            continue
        nonres = pw[one]
        T = np.zeros((TAU, TAU), dtype=np.uint64)
        for k, ext in enumerate(idx):                       # internal exponent k (Y^k) sits at external index idx[k]
            T[ext, k] = 1
        return T, nonres
    raise CalibrationError("ext_mul_tensor: no unit vector generates the basis: not a permuted binomial basis (lf_set_ext_basis by hand)")


def load_probe(src):
    """src: a path, JSON text or the parsed object of tools/probe_stark_rings.rs -> Calibration (CalibrationError if it is inconsistent)"""
    if isinstance(src, dict):
        d = src
    else:
        text = open(src).read() if isinstance(src, (str, bytes)) and os.path.exists(src) else src
        if isinstance(text, bytes):
            text = text.decode()
        # the probe prints Rust's Debug form: tuples come out as (a, b, c) -- the only parentheses in the object
        d = json.loads(text.replace("(", "[").replace(")", "]"))
    for key in ("nonres", "y", "crt_of_monomials", "digit_cases"):
        if key not in d:
            raise CalibrationError(f"probe output has no {key!r}")
    nonres = int(d["nonres"]) % P
    y_ext = np.array([[int(v) % P for v in s] for s in d["y"]], dtype=object)
    if y_ext.shape != (SLOTS, TAU):
        raise CalibrationError("probe 'y' must be 8 slots of 3 coordinates")
    # the extension field's basis: identity unless the crate's own table says otherwise
    T = np.eye(TAU, dtype=np.uint64)
    if "ext_mul_tensor_goldilocks_fq3" in d:
        T, nr_t = _ext_basis_from_tensor(d["ext_mul_tensor_goldilocks_fq3"], nonres)
        if nr_t != nonres:
            raise CalibrationError(f"probe 'nonres' ({nonres}) is not the cube of the generator its multiplication table shows ({nr_t})")
    Ti = [[int(T[j, i]) for j in range(TAU)] for i in range(TAU)]     # a permutation: the inverse is the transpose
    to_int = lambda v: [sum(Ti[i][j] * int(v[j]) for j in range(TAU)) % P for i in range(TAU)]
    to_ext = lambda v: [sum(int(T[i, j]) * v[j] for j in range(TAU)) % P for i in range(TAU)]
    y_int = [to_int(y_ext[k]) for k in range(SLOTS)]
    # cross-check: the crate's CRT of the unit monomials must be slot_k(X^j) = y_k^j in F_p[Y]/(Y^3 - nonres)
    crt = d["crt_of_monomials"]
    if len(crt) != RE or any(len(r) != RE for r in crt):
        raise CalibrationError("probe 'crt_of_monomials' must be 24 rows of 24 words")
    for k in range(SLOTS):
        pw = [1, 0, 0]
        for j in range(RE):
            if [int(x) % P for x in crt[j][TAU * k:TAU * k + TAU]] != to_ext(pw):
                raise CalibrationError(f"the crate's CRT of X^{j} in slot {k} is not y_k^{j}: its CRT is not the evaluation map of the (nonres, y) parametrisation "
                                       "-- use the general form (dense matrix + structure tensor: lfo_set_ring_general on the oracle; the product needs a new data API)")
            pw = _binomial_mul(pw, y_int[k], nonres)
    # the digit rule: whichever of the two rules reproduces every probed case
    modes = []
    for mode in (0, 1):
        ok = True
        for v, d16, d2 in d["digit_cases"]:
            sgn = lambda w: int(w) if int(w) <= (P - 1) // 2 else int(w) - P
            if [sgn(w) for w in d16] != balanced_digits(int(v), 1 << 16, 4, mode) or [sgn(w) for w in d2] != balanced_digits(int(v), 2, 16, mode):
                ok = False
                break
        if ok:
            modes.append(mode)
    if not modes:
        raise CalibrationError("probe 'digit_cases' match neither digit rule (lf_set_digit_mode 0 / 1): a third rule needs code, not data")
    ser_ok = True
    if "serialized_element" in d:
        se = d["serialized_element"]
        flat = b"".join(int(w).to_bytes(8, "little") for w in se["flat_words"])
        ser_ok = int(se["len"]) == len(flat) and bytes(se["bytes"]) == flat
    extra = {k: v for k, v in d.items() if k.startswith(("babybear", "frog", "ext_mul_tensor_babybear"))}
    if len(modes) > 1:
        import warnings
        warnings.warn("probe 'digit_cases' do not separate the two digit rules (no tie or negative-remainder case among them): lf_set_digit_mode 0 is taken "
                      "UNVERIFIED -- add a value of the form k B + B / 2 to tools/probe_stark_rings.rs", stacklevel=2)
    return Calibration(nonres=nonres, y=np.array(y_int, dtype=np.uint64), ext_basis=T, digit_mode=modes[0], serialized_words_le=ser_ok, extra=extra,
                       digit_mode_ambiguous=len(modes) > 1)
