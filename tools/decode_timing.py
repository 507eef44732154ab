#!/usr/bin/env python
"""GPU batch decode (fpng_amd_decode_batch: host PNG bytes -> device pixels, upload included) next to the drop-in's CPU decoder."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["FPNG_AMD_DECODE_CPU"] = "1"  # `CPU decoder` below means the drop-in's CPU tier (its GPU tier has its own tool: tools/dropin_decode_timing.py)
import numpy as np, torch, fpng_amd, dropin, real_image
enc = fpng_amd.Encoder(device=0)
imgs = real_image.variants(real_image.rgb_pixels(dropin.decode))
cases = [("photo 2748x4048 RGB x 8", [torch.from_numpy(imgs["rgb_t4"]).cuda()] * 8),
         ("4K RGBA grad x 16", [torch.from_numpy(fpng_amd.synth_image("grad", 3840, 2160, 4, seed=12345 + i)).cuda() for i in range(16)]),
         ("8K RGBA grad x 4", [torch.from_numpy(fpng_amd.synth_image("grad", 7680, 4320, 4, seed=12345 + i)).cuda() for i in range(4)]),
         ("8K RGBA gradient WITHOUT noise x 4 (seed 0: a periodic stream, dozens of rounds)", [torch.from_numpy(fpng_amd.synth_image("grad", 7680, 4320, 4, seed=0)).cuda()] * 4),
         ("1080p RGB grad x 64", [torch.from_numpy(fpng_amd.synth_image("grad", 1920, 1080, 3, seed=12345 + i)).cuda() for i in range(64)])]
for name, ts in cases:
    for flags in (0, 1):
        pngs, _ = enc.encode_tensors(ts, flags)
        mp = sum(t.shape[0] * t.shape[1] for t in ts) / 1e6
        dims = [(t.shape[1], t.shape[0]) for t in ts]
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter(); got = enc.decode_batch(pngs, 4, dims); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        assert all(st == 0 for st, _, _ in got)
        t0 = time.perf_counter(); st, px, w, h, c = dropin.decode(pngs[0], 4); cpu = time.perf_counter() - t0
        assert st == 0 and np.array_equal(got[0][1].cpu().numpy().reshape(-1), px)
        print(f"{name} flags={flags}: GPU batch {best*1e3:7.2f} ms = {mp/best/1e3:6.2f} GP/s ({sum(len(p) for p in pngs)/1e6:.0f} MB of PNG in) | CPU decoder, one file: {cpu*1e3:7.1f} ms = {ts[0].shape[0]*ts[0].shape[1]/1e6/cpu:6.1f} MP/s")
