#!/usr/bin/env python
"""Damage campaign against the GPU decoder on content whose token streams are periodic or flat (where the phase maps and the
gathered correction steps work) and on the photograph: every file -- whole, truncated, bits flipped all over the stream, header
edits -- through fpng_amd_decode_batch (host-resident) and fpng_amd_decode_batch_device, status AND pixels against the REFERENCE's
decoder (oracle/_ref).  FPNG_AMD_DECODE_UNDECIDED counts as a difference unless the reference fails too.
    python tools/gpu_decode_damage_campaign.py [copies per file]  ->  a summary on stdout"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, fpng_amd, dropin, real_image, ui_images
from cpu_ref import oracle, ref
from test_decode_model import periodic_images
n_copies = int(sys.argv[1]) if len(sys.argv) > 1 else 60
enc = fpng_amd.Encoder(device=0)
rng = np.random.default_rng(2026)
bases = []
for name, img, w, h, c in periodic_images():
    bases.append((name, img, w, h, c))
for name, (img, w, h, c) in sorted(ui_images.all_images().items()):
    if w == 1920:
        bases.append((name, np.asarray(img).reshape(-1), w, h, c))
photo = real_image.variants(real_image.rgb_pixels(dropin.decode))["rgb"]
bases.append(("photo", photo.reshape(-1), photo.shape[1], photo.shape[0], 3))
t0 = time.time()
tot = dict(files=0, ok=0, rejected=0, undecided=0, diff=0)
for name, img, w, h, c in bases:
    for flags in (0, 1):
        png = oracle().encode(img, w, h, c, flags)
        files = [png]
        for _ in range(n_copies):
            kind = int(rng.integers(0, 4))
            d = bytearray(png)
            if kind == 0:    # truncate
                d = d[: int(rng.integers(60, len(d)))]
            elif kind == 1:  # 1-3 flipped bits anywhere in the pixel stream
                for _ in range(int(rng.integers(1, 4))):
                    d[int(rng.integers(120, len(d)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 2:  # block header / code length region
                d[int(rng.integers(58, 140))] = int(rng.integers(0, 256))
            else:            # a random byte
                d[int(rng.integers(58, len(d)))] = int(rng.integers(0, 256))
            files.append(bytes(d))
        desired = 4 if (len(files) + flags) & 1 else 3
        want = [ref().decode(f, desired) for f in files]
        for device in (False, True):
            if device:
                dev = [torch.frombuffer(bytearray(f), dtype=torch.uint8).cuda() for f in files]
                got = enc.decode_device(dev, desired, [(w, h)] * len(files))
            else:
                got = enc.decode_batch(files, desired)
            for (st, px, cf), (rst, rpx, rw, rh, rc) in zip(got, want):
                tot["files"] += 1
                if st == 64:
                    tot["undecided"] += 1
                    tot["diff"] += rst == 0
                elif st != rst:
                    tot["diff"] += 1
                elif st == 0:
                    same = np.array_equal(px.cpu().numpy().reshape(-1), np.asarray(rpx)[: rw * rh * desired])
                    tot["ok"] += same
                    tot["diff"] += not same
                else:
                    tot["rejected"] += 1
    print(f"{name}: running totals {tot} after {time.time() - t0:.0f} s", flush=True)
print("RESULT", tot, "PASS" if tot["diff"] == 0 else "FAIL")
