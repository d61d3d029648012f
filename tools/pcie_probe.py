"""PCIe rates of the GPU box for the LatticeFold+ host-I/O figure: 134 MB witness-sized buffers, pageable / pinned, both directions; host scan and copy rates"""
import time
import numpy as np
import torch
n = (1 << 20) * 16
P = np.uint64(15912092521325583641)
a = (np.arange(n, dtype=np.uint64) * np.uint64(7)) % P
d = torch.empty(n, dtype=torch.int64, device="cuda")
ta = torch.from_numpy(a.view(np.int64))
pin = torch.empty(n, dtype=torch.int64).pin_memory()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
mb = n * 8 / 1e6
for name, fn in (("H2D pageable", lambda: d.copy_(ta)), ("H2D pinned", lambda: d.copy_(pin, non_blocking=True)), ("D2H pageable", lambda: ta.copy_(d)),
                 ("D2H pinned", lambda: pin.copy_(d, non_blocking=True)), ("host copy pageable->pinned", lambda: pin.copy_(ta)),
                 ("host scan (numpy >= p)", lambda: (a >= P).any()), ("np.zeros + touch", lambda: np.zeros(n, dtype=np.uint64).sum())):
    s = t(fn)
    print(f"{name:28s} {s * 1e3:7.2f} ms  {mb / s / 1e3:6.1f} GB/s")
t0 = time.perf_counter(); torch.cuda.cudart().cudaHostRegister(a.ctypes.data, n * 8, 0); t1 = time.perf_counter()
print(f"hipHostRegister 134 MB       {(t1 - t0) * 1e3:7.2f} ms")
tr = torch.from_numpy(a.view(np.int64))
print(f"H2D registered               {t(lambda: d.copy_(tr, non_blocking=True)) * 1e3:7.2f} ms")
t0 = time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(a.ctypes.data); t1 = time.perf_counter()
print(f"hipHostUnregister            {(t1 - t0) * 1e3:7.2f} ms")
