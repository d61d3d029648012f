"""Time the host Poseidon permutation through the transcript ABI (no GPU needed): python tools/time_poseidon.py [goldilocks|babybear|frog]"""
import ctypes, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("LFHIP_LIB") or os.path.join(root, "latticefold_amd", "liblfhip.so"))
lib.lf_transcript_new_ring.restype = ctypes.c_void_p
lib.lf_transcript_new_ring.argtypes = [ctypes.c_int]
lib.lf_transcript_absorb_fq.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
ring = sys.argv[1] if len(sys.argv) > 1 else "goldilocks"
if ring == "frog":      # the LatticeFold+ transcript (lfp_protocol.cpp / lfp_poseidon_simd.cc): 16 coefficients per ring element, 20 words per permutation
    lib.lfplus_transcript_new.restype = ctypes.c_void_p
    lib.lfplus_transcript_absorb.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    t = lib.lfplus_transcript_new()
    x = np.arange(16 * 50000, dtype=np.uint64)
    lib.lfplus_transcript_absorb(t, x.ctypes.data, 1000)
    best = 1e9
    for _ in range(int(os.environ.get('REPS', '3'))):
        t0 = time.perf_counter(); lib.lfplus_transcript_absorb(t, x.ctypes.data, 50000); best = min(best, time.perf_counter() - t0)
    print(ring, "scalar" if os.environ.get("LFPLUS_POSEIDON_SCALAR") else "simd=%d" % lib.lfplus_poseidon_simd(), "%.3f us/perm" % (best / 40000 * 1e6))
    sys.exit(0)
t = lib.lf_transcript_new_ring(1 if ring == "babybear" else 0)
x = np.arange(20 * 50000, dtype=np.uint64)
lib.lf_transcript_absorb_fq(t, x.ctypes.data, 20000)      # warm up
best = 1e9
for _ in range(int(os.environ.get('REPS', '3'))):
    t0 = time.perf_counter(); lib.lf_transcript_absorb_fq(t, x.ctypes.data, len(x)); best = min(best, time.perf_counter() - t0)
print(ring, "scalar" if os.environ.get("LF_POSEIDON_SCALAR") else "simd", "%.3f us/perm" % (best / 50000 * 1e6))
