#!/usr/bin/env python
"""Timing of the encode chain without any check of the files (for the timing-only builds of encode_direct_kernel, fpng_amd/build.py
--variant abl_*): K back-to-back submissions of B frames, then the kernels' own times.   python tools/direct_timing.py [WxHxC] [B] [flags]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, fpng_amd
w, h, c = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "7680x4320x4").split("x"))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
imgs = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, c, seed=12345 + i)).cuda() for i in range(B)]
cap = fpng_amd.max_encoded_size(w, h, c) + 64
outs = [[torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(B)] for _ in range(4)]
enc = fpng_amd.Encoder(device=0, stream="own")
for i in range(60):
    enc.submit(imgs, outs[i & 3], flags)
enc.finish(B)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20):
        enc.submit(imgs, outs[i & 3], flags)
    enc.finish(B); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
enc.set_profiling(True)
ph = np.zeros(8)
for _ in range(5):
    enc.submit(imgs, outs[0], flags); enc.finish(B); ph += np.array(enc.last_phase_ms())
print("%-28s %dx%dx%d x %d flags %d: %.4f ms/step = %6.1f GP/s | " % (os.path.basename(os.environ.get("FPNG_AMD_LIB", "product")) + " " + " ".join(f"{k[9:]}={v}" for k, v in os.environ.items() if k.startswith("FPNG_AMD_") and k != "FPNG_AMD_LIB"), w, h, c, B, flags, best * 1e3, B * w * h / best / 1e9)
      + " ".join("%s %.4f" % (n, v / 5) for n, v in zip(enc.phase_names(), ph) if v / 5 > 0.002), flush=True)
enc.close()
