#!/usr/bin/env python
"""PCIe-inclusive timing of the host-buffer entry points (what a user of the reference's API pays): host pixels in, host PNG
out.  1-pass frames of 16 MiB of pixels and more are streamed through the GPU in row bands, pageable or page-locked
(fpng_amd_pin_host_memory) alike; FPNG_AMD_HOST_BANDS=n forces n bands (1 = one direction at a time; read once per process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, fpng_amd
import dropin
enc = fpng_amd.Encoder(device=0, stream="own")
print("FPNG_AMD_HOST_BANDS =", os.environ.get("FPNG_AMD_HOST_BANDS", "(by size)"))


def best_of(fn, n=7):
    fn()
    b = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); b = min(b, time.perf_counter() - t0)
    return b, r


for (w, h, c) in [(1920, 1080, 3), (3840, 2160, 4), (7680, 4320, 4)]:
    for flags in (0, 1):
        img = fpng_amd.synth_image("grad", w, h, c)
        mp = w * h / 1e6
        out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
        res = {}
        res["C ABI pageable"], n = best_of(lambda: enc.encode_host_into(img, w, h, c, out, flags))
        res["fpng:: pageable, reused vector"], nd = dropin.time_encode(img, w, h, c, flags, reps=8, reuse=True)
        res["fpng:: pageable, fresh vector"], _ = dropin.time_encode(img, w, h, c, flags, reps=4, reuse=False)
        fpng_amd.pin_host_memory(img)
        res["C ABI pixels page-locked"], n2 = best_of(lambda: enc.encode_host_into(img, w, h, c, out, flags))
        res["fpng:: pixels page-locked, reused vector"], _ = dropin.time_encode(img, w, h, c, flags, reps=8, reuse=True)
        fpng_amd.pin_host_memory(out)
        res["C ABI pixels and output page-locked"], n3 = best_of(lambda: enc.encode_host_into(img, w, h, c, out, flags))
        fpng_amd.unpin_host_memory(out)
        fpng_amd.unpin_host_memory(img)
        assert n == nd == n2 == n3
        print(f"{w}x{h}x{c} flags={flags} png {n} B: " + " | ".join(f"{k} {v*1e3:.2f} ms ({mp/v/1e3:.1f} GP/s)" for k, v in res.items()))
