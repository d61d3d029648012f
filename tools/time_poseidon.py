"""Time the host Poseidon permutation through the transcript ABI (no GPU needed): python tools/time_poseidon.py [goldilocks|babybear]"""
import ctypes, os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "latticefold_amd", "liblfhip.so"))
lib.lf_transcript_new_ring.restype = ctypes.c_void_p
lib.lf_transcript_new_ring.argtypes = [ctypes.c_int]
lib.lf_transcript_absorb_fq.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
ring = sys.argv[1] if len(sys.argv) > 1 else "goldilocks"
t = lib.lf_transcript_new_ring(1 if ring == "babybear" else 0)
x = np.arange(20 * 50000, dtype=np.uint64)
lib.lf_transcript_absorb_fq(t, x.ctypes.data, 20000)      # warm up
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); lib.lf_transcript_absorb_fq(t, x.ctypes.data, len(x)); best = min(best, time.perf_counter() - t0)
print(ring, "scalar" if os.environ.get("LF_POSEIDON_SCALAR") else "simd", "%.3f us/perm" % (best / 50000 * 1e6))
