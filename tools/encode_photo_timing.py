#!/usr/bin/env python
"""Encode throughput on PHOTOGRAPHIC content (bench.py's workloads are synthetic): 8 x the reference's photograph tiled 4 x 4
(2748 x 4048 RGB, 11 MP) and its RGBA variant, device-resident, K steps enqueued back to back; per-kernel HIP-event times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, fpng_amd, dropin, real_image
imgs = real_image.variants(real_image.rgb_pixels(dropin.decode))
rgb = imgs["rgb_t4"]
rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
enc = fpng_amd.Encoder(device=0, stream="own")
for name, im in (("photo 11 MP RGB x 8", rgb), ("photo 11 MP RGBA x 8", rgba)):
    h, w, c = im.shape
    ts = [torch.from_numpy(np.ascontiguousarray(np.roll(im, 17 * i, axis=1))).cuda() for i in range(8)]
    cap = fpng_amd.max_encoded_size(w, h, c) + 64
    out_sets = [[torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(8)] for _ in range(4)]
    batches = [enc.make_batch(ts, o) for o in out_sets]
    for flags in (0, 1):
        for i in range(30):
            enc.submit(batches[i & 3], None, flags)
        enc.finish(8)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(20):
                enc.submit(batches[i & 3], None, flags)
            res = enc.finish(8)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
        enc.set_profiling(True); enc.submit(ts, out_sets[0], flags); enc.finish(8); ph = dict(zip(enc.phase_names(), enc.last_phase_ms())); enc.set_profiling(False)
        png = sum(r[0] for r in res)
        print(f"{name} flags={flags}: {best*1e3:.3f} ms per step = {8*w*h/best/1e9:.1f} GP/s ({8*w*h*c/1e6:.0f} MB of pixels -> {png/1e6:.0f} MB of PNG; "
              f"{(8*w*h*c+png)/best/1e12:.2f} TB/s algorithmic); " + " ".join(f"{k} {v:.3f}" for k, v in ph.items() if v), flush=True)
