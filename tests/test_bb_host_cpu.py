"""CPU-side checks of the BabyBear host code shipped in liblfhip.so (Poseidon + transcript run on the host, no GPU
needed): product vs oracle (liblfo_bb.so) and vs the reference's BabyBear KATs."""
import numpy as np

import lfo_bb as lfo
from latticefold_amd import api

PB = 15 * 2**27 + 1


def test_poseidon_params_and_sparse_equals_plain(kats):
    k = kats["poseidon_babybear_params"]
    ark, mds = api.poseidon_params("babybear")
    assert ark[:4].tolist() == k["ark_first"] and mds[-4:].tolist() == k["mds_last"]
    assert sum((i + 1) * int(v) for i, v in enumerate(ark)) % PB == k["ark_checksum"]
    assert sum((i + 1) * int(v) for i, v in enumerate(mds)) % PB == k["mds_checksum"]
    rng = np.random.default_rng(5)
    for _ in range(50):
        st = rng.integers(0, PB, size=24, dtype=np.uint64)
        a = api.poseidon_permute(st, plain=False, ring="babybear")
        b = api.poseidon_permute(st, plain=True, ring="babybear")
        o = st.copy()
        lfo.lib().lfo_poseidon_permute(lfo._p64(o))
        assert (a == b).all() and (a == o).all()


def test_simd_permutation_paths_agree():
    """transcript path (AVX-512 IFMA or AVX2 lanes, chosen at run time) vs scalar sparse factorisation vs textbook loop,
    including edge words; and the same digest from subprocesses that force the AVX2 and the scalar path."""
    import os, subprocess, sys
    edge = [0, 1, PB - 1, PB - 2, (PB - 1) // 2, (PB + 1) // 2, 2**27, 2**31 - 1 - PB + PB - 2**27]
    rng = np.random.default_rng(12)
    for it in range(300):
        st = (np.array([edge[(it * 5 + i * 3 + (i * it) % 7) % 8] % PB for i in range(24)], dtype=np.uint64) if it < 40
              else rng.integers(0, PB, size=24, dtype=np.uint64))
        a = api.poseidon_permute(st, 0, "babybear")
        assert (a == api.poseidon_permute(st, 2, "babybear")).all() and (a < PB).all()
        if it < 80:
            assert (a == api.poseidon_permute(st, 1, "babybear")).all()
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from latticefold_amd import api; "
            "s = np.arange(24, dtype=np.uint64) * np.uint64(77777);\n"
            "for _ in range(5): s = api.poseidon_permute(s, 0, 'babybear')\n"
            "print(' '.join(str(int(v)) for v in s))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for env in ({}, {"LF_POSEIDON_AVX2": "1"}, {"LF_POSEIDON_SCALAR": "1"}):
        e = dict(os.environ); e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.strip())
    assert outs[0] == outs[1] == outs[2] and len(outs[0].split()) == 24


def test_simd_permutation_chain_matches_scalar():
    """3000 chained permutations, transcript path (AVX-512 lanes with the collapsed partial rounds, bb_poseidon_avx512.cc) against the scalar sparse form"""
    st = (np.arange(24, dtype=np.uint64) * np.uint64(123456789)) % np.uint64(PB)
    ref = st.copy()
    for i in range(3000):
        st = api.poseidon_permute(st, 0, "babybear")
        ref = api.poseidon_permute(ref, 2, "babybear")
        if i % 250 == 0:
            assert (st == ref).all(), i
    assert (st == ref).all() and (st < np.uint64(PB)).all()


def test_transcript_matches_oracle(kats):
    t, o = api.PoseidonTranscript(ring="babybear"), lfo.Transcript()
    rng = np.random.default_rng(9)
    for step in range(40):
        kind = step % 4
        if kind == 0:
            x = rng.integers(0, PB, size=int(rng.integers(1, 60)), dtype=np.uint64)
            t.absorb_fq(x); o.absorb_fq(x)
        elif kind == 1:
            e = rng.integers(0, PB, size=(int(rng.integers(1, 4)), 72), dtype=np.uint64)
            t.absorb_slice(e); o.absorb_ring(e)
        elif kind == 2:
            assert (t.get_challenge() == o.challenge()).all()
        else:
            a, b = t.get_short_challenge(), o.short_challenge()
            assert (a == b).all() and not a[24:].any()
    c = t.clone()
    assert (c.get_challenge() == t.get_challenge()).all()


def test_small_challenge_decoder_kat(kats):
    """BabyBearChallengeSet KAT (rings/babybear.rs:77-114) through the oracle decoder; the product decoder is the same
    6-bit unpacking and is compared with the oracle on squeezed bytes in test_transcript_matches_oracle"""
    k = kats["babybear_small_challenge_from_bytes"]
    assert lfo.short_challenge_from_bytes(k["bytes"])[:24].tolist() == k["coeffs"]
