#!/usr/bin/env python
"""What can HBM do with the pipeline's OWN traffic mix?  The 8 x 8K step moves 1.62 GB of reads and 0.94 GB of writes
(profiles/r03_8k_pmc_traffic.txt).  Stream exactly that with the plain calibration kernels (16-byte lanes, nothing else to do),
reads and writes on two streams at once, and compare with the step's 0.47 ms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import torch
from stream_probe import stream as probe  # (tools/probes/stream_probe.hip; runs on torch's current stream)
R, W = 1_621_500_000 // 16 * 16, 941_900_000 // 16 * 16
rbuf = torch.zeros(R, dtype=torch.uint8, device="cuda")
wbuf = torch.zeros(W, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(do_r, do_w, reps=20):
    for k in range(3 + reps):
        if k == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        if do_r:
            with torch.cuda.stream(s1): probe(rbuf, 0, 16)
        if do_w:
            with torch.cuda.stream(s2): probe(wbuf, 1, 16)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
tr, tw, tb = run(1, 0), run(0, 1), run(1, 1)
print(f"reads alone  {R/1e9:.3f} GB: {tr:.3f} ms = {R/tr/1e9:.2f} TB/s")
print(f"writes alone {W/1e9:.3f} GB: {tw:.3f} ms = {W/tw/1e9:.2f} TB/s")
print(f"both at once {(R+W)/1e9:.3f} GB: {tb:.3f} ms = {(R+W)/tb/1e9:.2f} TB/s   (the encoder's step moves the same bytes in ~0.47 ms)")
# cross-check with the framework's own kernels (fill = write only, int32 sum = read only, copy = 1:1)
def t_of(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
w32, r32 = wbuf.view(torch.int32), rbuf.view(torch.int32)
dst = torch.empty_like(wbuf)
t = t_of(lambda: w32.fill_(7)); print(f"torch fill_  {W/1e9:.3f} GB written: {t:.3f} ms = {W/t/1e9:.2f} TB/s")
t = t_of(lambda: r32.sum()); print(f"torch sum    {R/1e9:.3f} GB read:    {t:.3f} ms = {R/t/1e9:.2f} TB/s")
t = t_of(lambda: dst.copy_(wbuf)); print(f"torch copy_  {W/1e9:.3f} GB read + {W/1e9:.3f} GB written: {t:.3f} ms = {2*W/t/1e9:.2f} TB/s")
def both():
    with torch.cuda.stream(s1): r32.sum()
    with torch.cuda.stream(s2): w32.fill_(7)
t = t_of(both); print(f"torch sum || fill_ ({(R+W)/1e9:.3f} GB): {t:.3f} ms = {(R+W)/t/1e9:.2f} TB/s")
# is the fill's rate a property of constant data or of the store shape?
N32 = w32.numel()
t = t_of(lambda: torch.arange(N32, out=w32, dtype=torch.int32)); print(f"torch arange(out=) {W/1e9:.3f} GB written: {t:.3f} ms = {W/t/1e9:.2f} TB/s")
def w4():
    probe(wbuf, 1, 4)
t = t_of(w4); print(f"own write kernel, 4-byte lanes: {t:.3f} ms = {W/t/1e9:.2f} TB/s")
def w16():
    probe(wbuf, 1, 16)
t = t_of(w16); print(f"own write kernel, 16-byte lanes: {t:.3f} ms = {W/t/1e9:.2f} TB/s")
