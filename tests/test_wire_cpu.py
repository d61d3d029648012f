"""Wire format of LFProof (lf_proof_serialize / lf_proof_deserialize, include/lfhip.h): host only, no GPU.
The layout follows the ark-serialize derive rules; the ring-element part is an assumption (see lf_wire.cpp), so these tests pin
the documented layout and the validation behaviour, not the reference's bytes."""
import struct

import numpy as np
import pytest

from latticefold_amd import api
from latticefold_amd.workload import make_workload

MOD = {"goldilocks": 0xFFFFFFFF00000001, "babybear": 15 * 2**27 + 1}


def rand_proof(wl, seed):
    prm = api._wl_params(wl)
    n = api._lib().lf_proof_len_ring(api.C.byref(prm), api.RING_IDS[wl.ring])
    rng = np.random.default_rng(seed)
    return rng.integers(0, MOD[wl.ring], size=(n, api.RING_WORDS[wl.ring]), dtype=np.uint64)


@pytest.mark.parametrize("name", ["T8", "G5", "B6"])
def test_round_trip_and_layout(name):
    wl = make_workload(name, 0)
    proof = rand_proof(wl, 1)
    data = api.proof_to_bytes(wl, proof)
    re_, tau = api.RING_WORDS[wl.ring], (3 if wl.ring == "goldilocks" else 9)
    # size: one u64 per Vec + 8 bytes per base-field word
    nvec = (1 + wl.s) + 2 + 2 * 4 * (1 + wl.K) + (1 + wl.s) + 2 * (1 + 2 * wl.K)
    assert len(data) == 8 * nvec + 8 * proof.size
    # LinearizationProof starts with the sumcheck Vec<ProverMsg>: s messages of d + 2 evaluations
    assert struct.unpack_from("<Q", data, 0)[0] == wl.s
    assert struct.unpack_from("<Q", data, 8)[0] == wl.d + 2
    first = np.frombuffer(data, dtype="<u8", count=re_, offset=16)
    assert (first == proof[0]).all()
    # after the s round messages: v (tau elements)
    off = 8 + wl.s * (8 + (wl.d + 2) * re_ * 8)
    assert struct.unpack_from("<Q", data, off)[0] == tau
    back = api.proof_from_bytes(wl, data)
    assert (back == proof).all()


def test_validation():
    wl = make_workload("T8", 0)
    proof = rand_proof(wl, 2)
    data = bytearray(api.proof_to_bytes(wl, proof))
    with pytest.raises(api.LfError):
        api.proof_from_bytes(wl, bytes(data[:-1]))                 # truncated
    with pytest.raises(api.LfError):
        api.proof_from_bytes(wl, bytes(data) + b"\0")              # trailing byte
    bad = bytearray(data); struct.pack_into("<Q", bad, 0, wl.s + 1)
    with pytest.raises(api.LfError):
        api.proof_from_bytes(wl, bytes(bad))                       # wrong Vec length
    bad = bytearray(data); struct.pack_into("<Q", bad, 16, MOD["goldilocks"])
    with pytest.raises(api.LfError):
        api.proof_from_bytes(wl, bytes(bad))                       # non-canonical field element
    nc = proof.copy(); nc[3, 5] = np.uint64(MOD["goldilocks"])
    with pytest.raises(api.LfError):
        api.proof_to_bytes(wl, nc)
    other = make_workload("T10", 0)                                 # bytes of one parameter set do not parse under another
    with pytest.raises(api.LfError):
        api.proof_from_bytes(other, bytes(data))


def test_oracle_proof_survives_the_wire_and_still_verifies():
    import lfo
    wl = make_workload("T8", 0)
    inst = lfo.Instance(wl)
    A = wl.ajtai_matrix()
    f_coeff = inst.witness_from_w_ccs(wl.w_ccs)
    cm = lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f_coeff))
    cccs = np.concatenate([cm, wl.x_ccs])
    acc, _ = inst.linearize(lfo.Transcript(), cccs, f_coeff)
    lc, _, proof = inst.fold_step(lfo.Transcript(), A, acc, f_coeff, cccs, f_coeff)
    back = api.proof_from_bytes(wl, api.proof_to_bytes(wl, proof))
    ok, lc_v, _ = api.NIFSVerifier.verify(wl, acc, cccs, back, api.PoseidonTranscript())
    assert ok and (lc_v == lc).all()
