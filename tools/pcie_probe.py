#!/usr/bin/env python
"""What do pageable / pinned host copies overlap with on this box?  (PCIe-inclusive design input)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

def t(fn, n=5):
    fn(); b = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3

IN, OUT = 133 << 20, 58 << 20
d_in = torch.empty(IN, dtype=torch.uint8, device="cuda"); d_out = torch.empty(OUT, dtype=torch.uint8, device="cuda")
for pinned in (False, True):
    h_in = torch.empty(IN, dtype=torch.uint8, pin_memory=pinned); h_out = torch.empty(OUT, dtype=torch.uint8, pin_memory=pinned)
    h_in.fill_(1); h_out.fill_(1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def up(chunks=1):
        with torch.cuda.stream(s1):
            n = IN // chunks
            for k in range(chunks):
                d_in[k * n:(k + 1) * n].copy_(h_in[k * n:(k + 1) * n], non_blocking=True)
            s1.synchronize()
    def down(chunks=1):
        with torch.cuda.stream(s2):
            n = OUT // chunks
            for k in range(chunks):
                h_out[k * n:(k + 1) * n].copy_(d_out[k * n:(k + 1) * n], non_blocking=True)
            s2.synchronize()
    def both(chunks=1):
        a = threading.Thread(target=up, args=(chunks,)); b = threading.Thread(target=down, args=(chunks,))
        a.start(); b.start(); a.join(); b.join()
    print(f"pinned={pinned}: H2D 133 MiB {t(up):.2f} ms | in 8 chunks {t(lambda: up(8)):.2f} | D2H 58 MiB {t(down):.2f} ms | in 8 chunks {t(lambda: down(8)):.2f} | "
          f"both at once (2 threads) {t(both):.2f} ms | both, 8 chunks each {t(lambda: both(8)):.2f} ms")
# cost of page-locking the caller's buffers per call
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
buf = np.ones(IN, dtype=np.uint8)
t0 = time.perf_counter(); rc = hip.hipHostRegister(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(IN), 0); t1 = time.perf_counter()
rc2 = hip.hipHostUnregister(ctypes.c_void_p(buf.ctypes.data)); t2 = time.perf_counter()
print(f"hipHostRegister 133 MiB: {1e3*(t1-t0):.2f} ms (rc {rc}), unregister {1e3*(t2-t1):.2f} ms (rc {rc2})")
