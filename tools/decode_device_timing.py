#!/usr/bin/env python
"""GPU decode of DEVICE-resident files (fpng_amd_decode_batch_device): the encoder's outputs decoded where they lie, per step of
n files: wall time of the call (host parse of the fetched heads + kernels + status read-back)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, fpng_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
only = sys.argv[2] if len(sys.argv) > 2 else ""
enc = fpng_amd.Encoder(device=0)
cases = [("8K RGBA grad x 8", "grad", 7680, 4320, 4, 8), ("4K RGBA grad x 16", "grad", 3840, 2160, 4, 16), ("1080p RGB grad x 64", "grad", 1920, 1080, 3, 64),
         ("512x512 RGB grad x 256", "grad", 512, 512, 3, 256), ("8K RGBA blocks x 8", "blocks", 7680, 4320, 4, 8),
         ("8K RGBA noise x 8", "noise", 7680, 4320, 4, 8), ("1080p RGB noise x 64", "noise", 1920, 1080, 3, 64), ("8K RGBA solid x 8", "solid", 7680, 4320, 4, 8)]
# a periodic token stream: stripes of five colours on a vertical ramp (every filtered row the same few bytes)
_pal = np.random.default_rng(3).integers(0, 256, (5, 4), dtype=np.uint8)
_stripes = np.ascontiguousarray((_pal[np.arange(7680) % 5][None] + (np.arange(4320)[:, None, None] * 7).astype(np.uint8)).astype(np.uint8))
cases.append(("8K RGBA stripes x 8", _stripes, 7680, 4320, 4, 8))
import dropin, real_image
_photo = real_image.variants(real_image.rgb_pixels(dropin.decode))["rgb_t4"]  # the reference's photograph, tiled 4 x 4: 2748 x 4048 RGB
cases.append(("photo 11 MP RGB x 8", _photo, _photo.shape[1], _photo.shape[0], 3, 8))
import ui_images
for uname, (uimg, uw, uh, uc) in sorted(ui_images.all_images().items()):
    if uw == 3840:
        cases.append((f"4K UI {uname} x 8", uimg, uw, uh, uc, 8))
for name, kind, w, h, c, n in cases:
    if only and only not in name:
        continue
    if isinstance(kind, str):
        ts = [torch.from_numpy(fpng_amd.synth_image(kind, w, h, c, seed=12345 + i)).cuda() for i in range(n)]
    else:  # (a fixed image: the same file n times)
        ts = [torch.from_numpy(kind).cuda()] * n
    for flags in (0, 1):
        pngs, _ = enc.encode_tensors(ts, flags)
        dev = [torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda() for p in pngs]
        outs = [torch.empty(w * h * c, dtype=torch.uint8, device="cuda") for _ in range(n)]
        dims = [(w, h)] * n
        best = 1e9
        db = enc.make_decode_batch(dev, c, dims, outs)  # (the descriptor array once: filling it costs Python ~3 us a file -- 0.6 ms of a 256-file step until round 6's last session)
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); enc.decode_device(db, results=False); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        got = db.results()
        if not os.environ.get("FPNG_TIMING_NOCHECK"):
            assert all(st == 0 for st, _, _ in got)
            assert all(torch.equal(px, t) for (st, px, _), t in zip(got, ts))
        ph = ""
        if os.environ.get("FPNG_TIMING_PHASES"):
            enc.set_profiling(True); enc.decode_device(db, results=False); ph = "; phases " + " ".join(f"{k} {v:.3f}" for k, v in enc.last_decode_phase_ms().items()); enc.set_profiling(False)
        mb = sum(len(p) for p in pngs) / 1e6
        print(f"{name} flags={flags}: {best*1e3:7.3f} ms per step = {n*w*h/best/1e9:7.2f} GP/s ({mb:.0f} MB of PNG -> {n*w*h*c/1e6:.0f} MB of pixels; "
              f"{(mb + n*w*h*c/1e6)/1e3/best:6.0f} GB/s algorithmic){ph}", flush=True)
