"""Streamed one-frame host path, pageable buffers: per-call times (FPNG_AMD_HOST_BANDS=8 streams from the very first call)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, fpng_amd, dropin
enc = fpng_amd.Encoder(device=0, stream="own")
for (w, h, c) in [(7680, 4320, 4), (3840, 2160, 4)]:
    img = fpng_amd.synth_image("grad", w, h, c)
    out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); n = enc.encode_host_into(img, w, h, c, out, 0); ts.append(round((time.perf_counter() - t0) * 1e3, 3))
    t_drop, nd = dropin.time_encode(img, w, h, c, 0, reps=8, reuse=True)
    print(f"{w}x{h}x{c} pageable: C ABI calls 1..6: {ts} ms (bands {enc.last_host_bands()}) | fpng:: drop-in, reused vector, best of 8: {t_drop*1e3:.3f} ms", n == nd)
