#!/usr/bin/env python
"""GPU-box check for the next round (not run yet: round 4's GPU minutes were spent): the encode kernels' stored-or-compressed decision
at the exact point where the reference's outcome flips -- images whose first K pixels are noise, K swept over the flip, 1-pass and
2-pass, as whole images and as 2-4 row bands through one GPU (tests/test_oracle.py and tests/test_sharded_cpu.py hold the CPU
checker and the band plan against the reference at the same points).  Prints one line per case; exit code 1 on any difference.
Becomes a -m gpu test once it has passed on a box."""
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
    enc.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
