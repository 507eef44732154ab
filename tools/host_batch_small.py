#!/usr/bin/env python
"""fpng_amd_encode_host_batch on MANY SMALL host frames (BASELINE config 3's frames coming from host memory): ms per frame, PCIe inclusive."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, fpng_amd
enc = fpng_amd.Encoder(device=0, stream="own")
for (w, h, c, n) in [(1920, 1080, 3, 256), (512, 512, 3, 1024), (3840, 2160, 4, 64), (7680, 4320, 4, 12)]:
    distinct = min(n, 32)
    imgs = [fpng_amd.synth_image("grad", w, h, c, seed=12345 + i) for i in range(distinct)]
    outs = [np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8) for _ in range(distinct)]
    frames = [imgs[i % distinct] for i in range(n)]
    fouts = [outs[i % distinct] for i in range(n)]
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); sizes = enc.encode_host_batch(frames, 0, outs=fouts); best = min(best, time.perf_counter() - t0)
    inb, outb = w * h * c, sizes[0]
    print(f"{n} x {w}x{h}x{c}: {best / n * 1e3:.4f} ms per frame = {w * h * n / best / 1e9:.2f} GP/s | upload alone at 53 GB/s {inb / 53e9 * 1e3:.4f} ms, download {outb / 53e9 * 1e3:.4f} ms", flush=True)
