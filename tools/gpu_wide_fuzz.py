#!/usr/bin/env python
"""Wide-row fuzz of the device-resident batch path against the CPU checker: rows of 257..4100 pixels built IN FILTERED SPACE from
segments -- runs of every interesting length (1, 2, the chunk limits 63/64 and 85/86, 255..260, ~1000), isolated one-pixel runs,
literal stretches -- at random alignment to the 256-pixel super-windows, so that runs cross super-window borders and the walk
hands over between its 4-pixels-per-lane phase and the per-pixel phase in mid-row.  All three flag values, both channel counts.
    python tools/gpu_wide_fuzz.py [cases per flag] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, fpng_amd
from cpu_ref import oracle

def wide_image(rng):
    c = int(rng.integers(3, 5))
    u = rng.random()
    if u < 0.3:
        w = int(rng.choice([256, 512, 768, 1024, 2048, 3840])) + int(rng.integers(-2, 3))
    else:
        w = int(rng.integers(257, 4101))
    h = int(rng.integers(1, 7))
    lens = [1, 1, 2, 2, 3, 62, 63, 64, 65, 84, 85, 86, 87, 126, 127, 128, 255, 256, 257, 258, 259, 260, 511, 512, 1000]
    p_lit = rng.random()
    F = np.zeros((h, w, c), dtype=np.uint8)
    for y in range(h):
        if y and rng.random() < 0.15:          # a row whose filtered bytes are all zero (identical to the row above)
            continue
        x = 0
        while x < w:
            if rng.random() < p_lit:           # literal stretch, sometimes with isolated pairs (the "sparse" tier)
                n = int(min(w - x, rng.integers(1, 700)))
                seg = rng.integers(0, 256, size=(n, c), dtype=np.uint8)
                if rng.random() < 0.5 and n > 4:
                    idx = rng.choice(np.arange(1, n), size=max(1, n // int(rng.integers(4, 40))), replace=False)
                    seg[idx] = seg[idx - 1]
                F[y, x:x + n] = seg
            else:                               # a run of one filtered value
                n = int(min(w - x, rng.choice(lens) if rng.random() < 0.7 else rng.integers(1, 1500)))
                F[y, x:x + n] = rng.integers(0, 256, size=(1, c), dtype=np.uint8) if rng.random() < 0.8 else 0
            x += n
    img = np.cumsum(F.astype(np.uint32), axis=0).astype(np.uint8)  # undo the Up filter: pixel = up + filtered (mod 256)
    return np.ascontiguousarray(img), w, h, c

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    enc = fpng_amd.Encoder(device=0)
    o = oracle()
    bad = 0
    for fl in (0, 1, 2):
        rng = np.random.default_rng(seed * 10 + fl)
        stored = 0
        for b in range(0, n, 200):
            cases = [wide_image(rng) for _ in range(min(200, n - b))]
            pngs, modes = enc.encode_tensors([torch.from_numpy(im).cuda() for im, _, _, _ in cases], fl)
            stored += sum(modes)
            for (im, w, h, c), p in zip(cases, pngs):
                if p != o.encode(im, w, h, c, fl):
                    bad += 1
                    print(f"DIFFERENT: flags={fl} {w}x{h}x{c}")
                    np.save(os.path.join(ROOT, "gpurun_out", f"wide_fuzz_bad_{fl}_{w}x{h}x{c}.npy"), im)
        print(f"flags={fl}: {n} wide images, {stored} stored, {'all identical to the checker' if not bad else str(bad) + ' DIFFERENT'}")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
