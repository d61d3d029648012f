"""CPU-side checks of the product library (no GPU needed): the C-ABI shared library loads and exports every
symbol include/lfhip.h declares, the host transcript inside the PRODUCT matches the reference KATs, and the
product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from latticefold_amd import api


def test_library_exports_every_declared_symbol():
    lib = api._lib()
    names = api.exported_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"liblfhip.so does not export {n}"
    from latticefold_amd import plus
    names = plus.exported_symbols()                      # include/lfplus.h (LatticeFold+ slice)
    assert len(names) >= 11
    for n in names:
        assert hasattr(plus._lib(), n), f"liblfhip.so does not export {n}"


def test_abi_version_matches_the_header():
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "lfhip.h")).read()
    assert api.abi_version() == int(re.search(r"#define LFHIP_ABI_VERSION (\d+)", hdr).group(1)) >= 5


def test_rust_binding_is_generated_from_the_headers():
    """INTEGRATION.md section 1 / bindings/latticefold-hip-sys/src/lib.rs are the mechanical image of the two headers: regenerating gives the committed
    text, every header symbol is declared exactly once, and the shared library exports each declared symbol"""
    import importlib.util
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("gen_rust_bindings", os.path.join(root, "tools", "gen_rust_bindings.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text, fns = gen.generate()
    assert open(gen.OUT).read() == text, "run tools/gen_rust_bindings.py --write"
    doc = open(gen.DOC).read()
    block = doc[doc.index(gen.BEGIN) + len(gen.BEGIN):doc.index(gen.END)]
    assert block.strip() == ("```rust\n" + text + "```").strip(), "INTEGRATION.md binding block is stale: run tools/gen_rust_bindings.py --write"
    from latticefold_amd import plus
    declared = re.findall(r"pub fn (\w+)\(", text)
    assert sorted(declared) == sorted(set(api.exported_symbols()) | set(plus.exported_symbols())) and len(declared) == len(set(declared))
    lib = api._lib()
    assert all(hasattr(lib, n) for n in declared)


def test_lfplus_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from latticefold_amd import plus
    with pytest.raises(plus.LfPlusError) as e:
        plus.PlusContext(0)
    assert e.value.code == plus.E_NO_DEVICE
    plus.scratch_trim()            # an empty scratch cache and no device: nothing to release, nothing to fail


def test_product_poseidon_params_and_kats(kats):
    k = kats["poseidon_goldilocks_params"]
    ark, mds = api.poseidon_params()
    P = api.P
    assert [int(x) for x in ark[:4]] == k["ark_first"] and [int(x) for x in mds[-4:]] == k["mds_last"]
    assert sum((i + 1) * int(v) for i, v in enumerate(ark)) % P == k["ark_checksum"]
    assert sum((i + 1) * int(v) for i, v in enumerate(mds)) % P == k["mds_checksum"]
    tr = api.PoseidonTranscript()
    tr.absorb_fq(kats["poseidon_big_challenge"]["absorb"])
    assert [int(x) for x in tr.get_challenge()] == kats["poseidon_big_challenge"]["expected_fq3"]
    tr = api.PoseidonTranscript()
    tr.absorb_fq(kats["poseidon_small_challenge"]["absorb"])
    assert [int(x) for x in tr.get_short_challenge()] == kats["poseidon_small_challenge"]["expected_coeffs"]


def test_product_transcript_matches_oracle_on_long_streams():
    """multi-block absorbs and squeeze-after-squeeze boundaries (31 consecutive short challenges cross the rate)."""
    import lfo
    rng = np.random.default_rng(7)
    a, b = api.PoseidonTranscript(), lfo.Transcript()
    for rep in range(3):
        x = rng.integers(0, 2**63, size=(5 + rep, 24), dtype=np.uint64)
        a.absorb_slice(x); b.absorb_ring(x)
        for _ in range(4):
            assert (a.get_challenge() == b.challenge()).all()
        for _ in range(31):
            assert (a.get_short_challenge() == b.short_challenge()).all()
    c = a.clone()
    assert (c.get_challenge() == a.get_challenge()).all()


def test_sparse_partial_rounds_equal_plain_permutation():
    import time
    import lfo
    rng = np.random.default_rng(3)
    for _ in range(50):
        st = rng.integers(0, 2**63, size=24, dtype=np.uint64)
        o = st.copy()
        lfo.lib().lfo_poseidon_permute(lfo._p64(o))
        assert (api.poseidon_permute(st, plain=True) == o).all()
        assert (api.poseidon_permute(st, plain=False) == o).all()
    assert (api.poseidon_permute(np.zeros(24, dtype=np.uint64)) == api.poseidon_permute(np.zeros(24, dtype=np.uint64), True)).all()


def test_simd_permutation_equals_scalar_and_plain():
    """plain = 0: transcript path (AVX-512 IFMA lanes when the CPU has them), 2: scalar sparse factorisation, 1: textbook loop."""
    P = 0xFFFFFFFF00000001
    edge = [0, 1, P - 1, P - 2, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 32, (P - 1) // 2]
    rng = np.random.default_rng(11)
    for it in range(400):
        if it < 40:
            st = np.array([edge[(it * 7 + i * 3 + (i * it) % 5) % 8] for i in range(24)], dtype=np.uint64)
        else:
            st = (rng.integers(0, 2**63, 24, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 24, dtype=np.uint64)) % np.uint64(P)
        a = api.poseidon_permute(st, 0)
        assert (a == api.poseidon_permute(st, 2)).all()
        if it < 80:
            assert (a == api.poseidon_permute(st, 1)).all()
        assert (a < np.uint64(P)).all()


def test_simd_permutation_chain_matches_scalar():
    """3000 chained permutations through the transcript's path (AVX-512 IFMA lanes when the CPU has them: elements as (u, v) pairs with a
    compare-free reduction, lf_poseidon_simd.cc) and through the scalar sparse form: every intermediate state feeds the next one, so 3000
    different non-canonical internal representations are exercised; compared every 250 steps and at the end"""
    P = api.P
    st = np.array([(i * 0x9E3779B97F4A7C15 + 12345) % P for i in range(24)], dtype=np.uint64)
    ref = st.copy()
    for i in range(3000):
        st = api.poseidon_permute(st, 0)
        ref = api.poseidon_permute(ref, 2)
        if i % 250 == 0:
            assert (st == ref).all(), i
    assert (st == ref).all() and (st < np.uint64(P)).all()


def test_scalar_transcript_env_gives_same_challenges():
    import subprocess, sys, os
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from latticefold_amd import api; t = api.PoseidonTranscript(); "
            "t.absorb(np.arange(24 * 50, dtype=np.uint64).reshape(50, 24)); print(' '.join(str(int(v)) for v in t.get_challenge()))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for env in ({}, {"LF_POSEIDON_SCALAR": "1"}):
        e = dict(os.environ); e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] == outs[1] and len(outs[0].split()) == 3


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.LfError) as e:
        api.Context(0)
    assert e.value.code == -2  # LF_ERR_HIP


def test_sizes_match_oracle_layout():
    import lfo
    from latticefold_amd.workload import make_workload
    for name in ("T8", "G5", "C2"):
        wl = make_workload(name) if name != "C2" else None
        if wl is None:
            continue
        inst = lfo.Instance(wl)
        p = api.Params(wl.s, wl.wit_len, wl.l, wl.L, wl.K, wl.b, wl.B, wl.kappa, wl.t, wl.q, wl.d)
        L = api._lib()
        assert L.lf_lcccs_len(C.byref(p)) == inst.lcccs_len
        assert L.lf_cccs_len(C.byref(p)) == inst.cccs_len
        assert L.lf_proof_len(C.byref(p)) == inst.proof_len


def test_ring_helpers_and_sizes():
    """ring-selection helpers of the ABI need no GPU: words / tau / modulus per ring and the flat sizes with v[tau]"""
    import ctypes as C
    from latticefold_amd import api
    L = api._lib()
    L.lf_ring_words.argtypes = [C.c_int]
    L.lf_ring_tau.argtypes = [C.c_int]
    assert (L.lf_ring_words(0), L.lf_ring_tau(0), L.lf_ring_modulus(0)) == (24, 3, 2**64 - 2**32 + 1)
    assert (L.lf_ring_words(1), L.lf_ring_tau(1), L.lf_ring_modulus(1)) == (72, 9, 15 * 2**27 + 1)
    assert L.lf_ring_words(7) == 0
    p = api.Params(10, 256, 1, 4, 16, 2, 1 << 16, 21, 3, 2, 2)
    for ring, tau in ((0, 3), (1, 9)):
        assert L.lf_lcccs_len_ring(C.byref(p), ring) == 10 + tau + 21 + 3 + 1 + 1
        assert L.lf_cccs_len_ring(C.byref(p), ring) == 21 + 1
        lin = 10 * 4 + tau + 3
        dec = 16 * (3 + tau + 2 + 21)
        fold = 10 * 5 + 2 * 16 * (tau + 3)
        assert L.lf_proof_len_ring(C.byref(p), ring) == lin + 2 * dec + fold
    assert L.lf_lcccs_len(C.byref(p)) == L.lf_lcccs_len_ring(C.byref(p), 0)
    assert L.lf_transcript_new_ring(5) is None
    L.lf_strerror.restype = C.c_char_p
    assert b"reject" in L.lf_strerror(-8)


def test_switch_table_is_current():
    """SWITCHES.md is the generated list of every environment switch the library reads (tools/list_switches.py): regenerating gives the committed text"""
    import importlib.util
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("list_switches", os.path.join(root, "tools", "list_switches.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert open(os.path.join(root, "SWITCHES.md")).read() == mod.render(), "run tools/list_switches.py --write"
    rows, _ = mod.collect()
    assert len(rows) > 30 and all(text for _d, text, _w in rows.values())


def test_transcript_squeeze_bytes_matches_the_short_challenge():
    """lf_transcript_squeeze_bytes = CryptographicSponge::squeeze_bytes of the reference's PoseidonSponge (7 / 3 usable bytes per field element): squeeze_bytes(18) on a
    clone gives the bytes whose 24 six-bit fields minus 32 are get_short_challenge's coefficients (rings/goldilocks.rs, rings/babybear.rs challenge sets; KAT-pinned)"""
    import ctypes as C
    lib = api._lib()
    lib.lf_transcript_squeeze_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t]
    lib.lf_transcript_squeeze_bytes.restype = None
    for ring, p, words in (("goldilocks", 0xFFFFFFFF00000001, 24), ("babybear", 15 * 2**27 + 1, 72)):
        t = api.PoseidonTranscript(ring=ring)
        t.absorb_slice(np.arange(3 * words, dtype=np.uint64).reshape(3, words) % np.uint64(p))
        t2 = t.clone()
        want = t.get_short_challenge()
        buf = (C.c_uint8 * 18)()
        lib.lf_transcript_squeeze_bytes(t2.h, buf, 18)
        bs = bytes(buf)
        got = []
        for g in range(6):
            w = bs[3 * g] | (bs[3 * g + 1] << 8) | (bs[3 * g + 2] << 16)
            got += [((w >> (6 * j)) & 63) - 32 for j in range(4)]
        assert [int(v) if int(v) < p // 2 else int(v) - p for v in want[:24]] == got
