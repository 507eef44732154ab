#!/usr/bin/env python
"""PCIe-inclusive timing of the host-buffer entry point (what the fpng:: drop-in pays): H2D + kernels + D2H."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, fpng_amd
enc = fpng_amd.Encoder(device=0)
for (w, h, c) in [(3840, 2160, 4), (7680, 4320, 4), (1920, 1080, 3)]:
    img = fpng_amd.synth_image("grad", w, h, c)
    enc.encode_host(img, w, h, c, 0)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); png = enc.encode_host(img, w, h, c, 0); best = min(best, time.perf_counter() - t0)
    print(f"{w}x{h}x{c}: {best*1e3:.2f} ms per frame  {w*h/best/1e6:.0f} MP/s  (png {len(png)} B, pageable host memory)")
