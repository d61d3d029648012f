"""The calibration path ("conventions are data"): tools/probe_stark_rings.rs prints the absent stark-rings crate's conventions, latticefold_amd/calibrate.py
loads that object and installs it through lf_set_ring_tables / lf_set_ext_basis / lf_set_digit_mode and the oracle's twins.  No Rust toolchain exists here, so the
probe's OUTPUT FORMAT is synthesised from the oracle under a convention that differs from the defaults (a permuted slot map, the floor digit rule) -- the day a
machine with the reference prints the real object, loading it is one call.  CPU part: the loader and the oracle; tests/test_gpu_calibration.py: the product."""
import json

import numpy as np
import pytest

import lfo
from latticefold_amd import calibrate
from latticefold_amd.calibrate import CalibrationError, balanced_digits, load_probe

P = calibrate.P
CASES = [0, 1, P - 1, 32767, 32768, 32769, P - 32768, P - 32769, (P - 1) // 2, (P + 1) // 2, 65535, 65536]


def _digits(v, base, digits):
    x = np.zeros((1, 24), dtype=np.uint64)
    x[0, 0] = np.uint64(v)
    return [int(w) for w in lfo.decompose(x, base, digits, 0).reshape(digits, 24)[:, 0]]


def synth_probe_text(nonres, y, mode, tensor_perm=None):
    """what tools/probe_stark_rings.rs would print on a crate with these conventions -- Rust Debug formatting included (tuples in parentheses, big integers of the
    tensor as strings); the oracle computes every entry.  tensor_perm: external index of the internal basis vector Y^k (a permuted / tower basis)"""
    lfo.set_ring(nonres, y)
    lfo.set_digit_mode(mode)
    perm = list(tensor_perm) if tensor_perm is not None else [0, 1, 2]
    ext = lambda w: [int(w[perm.index(e)]) for e in range(3)]               # internal coordinates -> the crate's order
    rows = []
    for j in range(24):
        e = np.zeros((1, 24), dtype=np.uint64)
        e[0, j] = 1
        w = [int(x) for x in lfo.crt(e).reshape(-1)]
        rows.append([c for k in range(8) for c in ext(w[3 * k:3 * k + 3])])
    cases = [(v, _digits(v, 1 << 16, 4), _digits(v, 2, 16)) for v in CASES]
    unit = lambda i: [int(k == i) for k in range(3)]
    tens = []
    for i in range(3):
        for j in range(3):
            a, b = np.array(unit(perm.index(i)), dtype=np.uint64), np.array(unit(perm.index(j)), dtype=np.uint64)
            o = np.zeros(3, dtype=np.uint64)
            lfo.lib().lfo_fq3_mul(lfo._p64(a), lfo._p64(b), lfo._p64(o))
            tens.append([str(v) for v in ext(o)])
    el = np.array([[1000 + i for i in range(24)]], dtype=np.uint64)
    flat = [int(x) for x in lfo.crt(el).reshape(-1)]
    flat = [c for k in range(8) for c in ext(flat[3 * k:3 * k + 3])]
    by = list(b"".join(w.to_bytes(8, "little") for w in flat))
    ys = [ext([int(v) for v in y[k]]) for k in range(8)]
    t = "{\"nonres\": %d,\n \"y\": %s,\n \"crt_of_monomials\": %s,\n" % (nonres, json.dumps(ys), json.dumps(rows))
    t += " \"digit_cases\": [%s],\n" % ", ".join("(%d, %s, %s)" % (v, json.dumps(a), json.dumps(b)) for v, a, b in cases)
    t += " \"ext_mul_tensor_goldilocks_fq3\": %s,\n" % json.dumps(tens)
    t += " \"frog_exp\": [[1, 0]],\n"
    t += " \"serialized_element\": {\"len\": %d, \"bytes\": %s, \"flat_words\": %s}}" % (len(by), json.dumps(by), json.dumps(flat))
    return t, rows


@pytest.fixture
def oracle_state():
    nr, y = lfo.get_ring()
    yield nr, y.copy()
    lfo.set_ring(nr, y)
    lfo.set_digit_mode(0)


def test_python_digit_rules_are_the_oracles(oracle_state):
    rng = np.random.default_rng(4)
    vals = CASES + [int(v) for v in rng.integers(0, P, size=200, dtype=np.uint64)] + [(1 << 47) + 32768, P - (1 << 31) - 32768, 98304, P - 98304]
    for mode in (0, 1):
        lfo.set_digit_mode(mode)
        for v in vals:
            sgn = lambda w: w if w <= (P - 1) // 2 else w - P
            assert [sgn(w) for w in _digits(v, 1 << 16, 4)] == balanced_digits(v, 1 << 16, 4, mode), (mode, v)
            assert [sgn(w) for w in _digits(v, 2, 16)] == balanced_digits(v, 2, 16, mode), (mode, v)
    assert balanced_digits(32768, 1 << 16, 4, 0) != balanced_digits(32768, 1 << 16, 4, 1)      # the probed cases do tell the rules apart


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("perm_seed", [None, 3])
def test_probe_output_round_trips_through_the_loader_and_switches_the_oracle(tmp_path, oracle_state, mode, perm_seed):
    nr0, y0 = oracle_state
    y = y0.copy()
    if perm_seed is not None:
        y = y[np.random.default_rng(perm_seed).permutation(8)]          # another slot order than the default: same roots, permuted
    text, rows = synth_probe_text(nr0, y, mode)
    path = tmp_path / "stark_rings_tables.json"
    path.write_text(text)
    lfo.set_ring(nr0, y0)                                               # back to the defaults: the calibration must move the oracle, not find it there
    lfo.set_digit_mode(0)
    cal = load_probe(str(path))
    assert cal.nonres == nr0 and (cal.y == y).all() and cal.digit_mode == mode and cal.serialized_words_le
    assert (cal.ext_basis == np.eye(3, dtype=np.uint64)).all() and "frog_exp" in cal.extra
    cal.apply_oracle(lfo)
    for j in (1, 5, 23):
        e = np.zeros((1, 24), dtype=np.uint64)
        e[0, j] = 1
        assert [int(x) for x in lfo.crt(e).reshape(-1)] == rows[j]
    sgn = lambda w: w if w <= (P - 1) // 2 else w - P
    assert [sgn(w) for w in _digits(32768, 1 << 16, 4)] == balanced_digits(32768, 1 << 16, 4, mode)
    assert load_probe(text).digit_mode == mode and load_probe(json.loads(text.replace("(", "[").replace(")", "]"))).nonres == nr0     # text and parsed forms too


def test_a_permuted_extension_basis_is_recovered_from_the_crates_own_table(oracle_state):
    nr0, y0 = oracle_state
    perm = [2, 0, 1]                                                    # Y^0 at external index 2, Y^1 at 0, Y^2 at 1
    text, _ = synth_probe_text(nr0, y0, 0, tensor_perm=perm)
    cal = load_probe(text)
    T = np.zeros((3, 3), dtype=np.uint64)
    for k, e in enumerate(perm):
        T[e, k] = 1
    assert (cal.ext_basis == T).all() and (cal.y == y0).all() and cal.nonres == nr0


def test_inconsistent_probe_output_is_refused(oracle_state):
    nr0, y0 = oracle_state
    text, _ = synth_probe_text(nr0, y0, 1)
    d = json.loads(text.replace("(", "[").replace(")", "]"))
    bad = json.loads(json.dumps(d))
    bad["crt_of_monomials"][7][4] = (bad["crt_of_monomials"][7][4] + 1) % P
    with pytest.raises(CalibrationError, match="CRT"):
        load_probe(bad)
    bad = json.loads(json.dumps(d))
    bad["digit_cases"][4][1][0] = 12345
    with pytest.raises(CalibrationError, match="digit"):
        load_probe(bad)
    bad = json.loads(json.dumps(d))
    bad["nonres"] = (nr0 + 1) % P
    with pytest.raises(CalibrationError):
        load_probe(bad)
    bad = json.loads(json.dumps(d))
    del bad["y"]
    with pytest.raises(CalibrationError, match="'y'"):
        load_probe(bad)
    bad = json.loads(json.dumps(d))
    bad["serialized_element"]["bytes"][3] ^= 1
    assert not load_probe(bad).serialized_words_le                      # reported, not fatal: the wire layout is lf_wire.cpp's business


def test_digit_cases_that_do_not_separate_the_rules_are_flagged(oracle_state):
    """a probe whose digit cases hold no tie / negative-remainder value fits both digit rules: the loader says so instead of silently taking rule 0"""
    nr0, y0 = oracle_state
    text, _ = synth_probe_text(nr0, y0, 0)
    d = json.loads(text.replace("(", "[").replace(")", "]"))
    assert not load_probe(d).digit_mode_ambiguous
    from latticefold_amd.calibrate import balanced_digits
    sgn = lambda x: x % P
    d["digit_cases"] = [[v, [sgn(x) for x in balanced_digits(v, 1 << 16, 4, 0)], [sgn(x) for x in balanced_digits(v, 2, 16, 0)]] for v in (0, 1, 5, 1000)
                        if balanced_digits(v, 1 << 16, 4, 0) == balanced_digits(v, 1 << 16, 4, 1) and balanced_digits(v, 2, 16, 0) == balanced_digits(v, 2, 16, 1)]
    assert d["digit_cases"]
    with pytest.warns(UserWarning, match="do not separate"):
        cal = load_probe(d)
    assert cal.digit_mode_ambiguous and cal.digit_mode == 0
