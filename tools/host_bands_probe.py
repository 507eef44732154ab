#!/usr/bin/env python
"""fpng_amd_encode_host (one call per frame, pageable buffers) for a given FPNG_AMD_HOST_BANDS (environment): ms per call, best of 12."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, fpng_amd
enc = fpng_amd.Encoder(device=0, stream="own")
res = []
for (w, h, c) in [(3840, 2160, 4), (1920, 1080, 3), (7680, 4320, 4), (2748, 4048, 3)]:
    img = fpng_amd.synth_image("grad", w, h, c)
    out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
    best = 1e9
    for _ in range(12):
        t0 = time.perf_counter(); enc.encode_host_into(img, w, h, c, out, 0); best = min(best, time.perf_counter() - t0)
    res.append(f"{w}x{h}x{c} {best * 1e3:.3f} ms ({enc.lib.fpng_amd_encoder_last_host_bands(enc.h)} bands)")
print("FPNG_AMD_HOST_BANDS=" + os.environ.get("FPNG_AMD_HOST_BANDS", "default") + ": " + " | ".join(res), flush=True)
