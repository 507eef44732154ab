#!/usr/bin/env python
"""GPU-box checks for the next round (not run yet: round 4's GPU minutes were spent when they were written).  Each is held on the
CPU already -- checker and band plans against the reference: tests/test_oracle.py, tests/test_sharded_cpu.py -- and becomes a
-m gpu test once it has passed on a box:
 1. the encode kernels' stored-or-compressed decision at the exact point where the reference's outcome flips: images whose first K
    pixels are noise, K swept over the flip, 1-pass and 2-pass;
 2. the device-side table builder (build_dynamic_kernel) on skewed histograms: filtered bytes with `levels` values whose counts
    grow like ratio^k -- from 13 levels on the optimal code is deeper than fpng's 12 bits and the length limiting decides -- and
    the GPU decoder on those files (dec_build_lut_kernel with 12-bit codes).
 3. token-edited MEGAPIXEL files (tests/token_mutator.py: LargeStream / mutate_large -- one local edit at a random place of a stream that
    spans many workgroups) through fpng_amd_decode_batch, fpng_amd_decode_batch_device and fpng::fpng_decode_memory (the streamed
    form from 8 MiB of IDAT on): status and pixels of the reference's decoder.
Prints one line per case; exit code 1 on any difference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fpng_amd  # noqa: E402
from cpu_ref import ref  # noqa: E402


def main():
    enc = fpng_amd.Encoder(device=0)
    rng = np.random.default_rng(2718)
    bad = 0
    stored = lambda png: (png[60] >> 1) & 3 == 0
    for (w, h, c) in ((64, 32, 4), (61, 17, 3), (256, 9, 4), (85, 30, 3), (33, 33, 4), (1024, 3, 3), (1920, 8, 4), (4096, 5, 3)):
        noise = rng.integers(0, 256, (w * h, c), dtype=np.uint8)
        for flags in (0, 1):
            def make(k):
                img = np.full((w * h, c), 77, dtype=np.uint8)
                img[:k] = noise[:k]
                return img.reshape(h, w, c)
            lo, hi = 0, w * h
            if not stored(ref().encode(make(hi), w, h, c, flags)):
                continue
            while hi - lo > 1:
                mid = (lo + hi) // 2
                lo, hi = (lo, mid) if stored(ref().encode(make(mid), w, h, c, flags)) else (mid, hi)
            ks = list(range(max(0, hi - 40), min(w * h, hi + 40) + 1))
            imgs = [make(k) for k in ks]
            pngs, _ = enc.encode_tensors([torch.from_numpy(i).cuda() for i in imgs], flags)
            n_bad = sum(bytes(p) != ref().encode(i, w, h, c, flags) for p, i in zip(pngs, imgs))
            bad += n_bad
            print(f"{w}x{h}x{c} flags {flags}: flip at K = {hi}, {len(ks)} images around it, {n_bad} differ from the reference", flush=True)
    from test_oracle import skewed_images
    imgs = skewed_images(np.random.default_rng(1618), max_bytes=3_000_000)
    for flags in (0, 1):
        pngs, _ = enc.encode_tensors([torch.from_numpy(i.reshape(h, w, c)).cuda() for i, w, h, c in imgs], flags)
        n_bad = sum(bytes(p) != ref().encode(i, w, h, c, flags) for p, (i, w, h, c) in zip(pngs, imgs))
        back = enc.decode_batch(pngs, 4)
        n_dec = sum(st != 0 or not np.array_equal(px.cpu().numpy()[:, :, :c].reshape(-1), i) for (st, px, _), (i, w, h, c) in zip(back, imgs))
        bad += n_bad + n_dec
        print(f"skewed histograms, flags {flags}: {len(imgs)} images, {n_bad} files differ from the reference, {n_dec} do not decode back", flush=True)
    import dropin
    import test_decode_model as M
    import token_mutator as TM
    import ui_images
    big = [(np.asarray(fpng_amd.synth_image("grad", 3840, 2160, 4)).reshape(-1), 3840, 2160, 4),   # (IDAT > 8 MiB: the drop-in streams it)
           (np.ascontiguousarray(ui_images.glyphs(1920, 1080, 3, seed=5)).reshape(-1), 1920, 1080, 3),
           (np.asarray(fpng_amd.synth_image("blocks", 2048, 1500, 3)).reshape(-1), 2048, 1500, 3)]
    for k, (img, w, h, c) in enumerate(big):
        s = TM.LargeStream(ref().encode(img, w, h, c, k % 2), M.plan, M.emul())
        files = [(name, f) for name, f in (TM.mutate_large(s, rng) for _ in range(24)) if f is not None]
        n_bad = 0
        for desired in (3, 4):
            judged = [ref().decode(f, desired) for _, f in files]
            for got in (enc.decode_batch([f for _, f in files], desired),):
                for (name, f), (cst, cpx, *_), (st, px, _) in zip(files, judged, got):
                    if st == 64:  # left to the CPU decoder: the drop-in's answer counts
                        st, dpx, *_ = dropin.decode(f, desired)
                        ok = st == cst and (cst != 0 or np.array_equal(np.asarray(dpx)[: w * h * desired], np.asarray(cpx)[: w * h * desired]))
                    else:
                        ok = st == cst and (cst != 0 or np.array_equal(px.cpu().numpy().reshape(-1), np.asarray(cpx)[: w * h * desired]))
                    n_bad += not ok
            for (name, f), (cst, cpx, *_) in zip(files, judged):  # the drop-in itself (GPU tier, streamed where the file is large)
                st, dpx, *_ = dropin.decode(f, desired)
                n_bad += not (st == cst and (cst != 0 or np.array_equal(np.asarray(dpx)[: w * h * desired], np.asarray(cpx)[: w * h * desired])))
        bad += n_bad
        print(f"token-edited {w}x{h}x{c}: {len(files)} files, {n_bad} answers differ from the reference's", flush=True)
    enc.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
