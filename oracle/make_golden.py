"""Generate tests/golden/* from the UNMODIFIED reference (oracle/_ref/libfpng_ref.so).

Run in the dev container (where /root/reference exists):   python oracle/make_golden.py
Outputs (committed):
  tests/golden/kat.json        size + sha256 of the reference encoder's output for the synthetic
                               images of SURVEY.md B.1/B.2 (flags 0, 1, 2)
  tests/golden/small/*.png     a handful of tiny complete reference outputs (byte fixtures)
  tests/golden/fuzz_cases.npz  120 small fuzz images (SURVEY.md B.3 recipe) with the reference's
                               output for flags 0 and 1
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cpu_ref import fuzz_image, ref  # noqa: E402
import fpng_amd  # noqa: E402  (only for the synthetic image generator, host code)

CASES = [
    (1, 1, 3, "solid"), (1, 1, 4, "solid"), (2, 1, 4, "solid"), (64, 1, 4, "solid"), (65, 1, 4, "solid"),
    (128, 2, 4, "solid"), (86, 1, 3, "solid"), (87, 1, 3, "solid"), (512, 512, 3, "grad"), (512, 512, 3, "blocks"),
    (512, 512, 4, "grad"), (100, 37, 3, "noise"), (100, 37, 4, "noise"),
    (1920, 1080, 3, "grad"), (1920, 1080, 3, "blocks"), (3840, 2160, 4, "grad"), (3840, 2160, 4, "blocks"),
    (3840, 2160, 4, "solid"), (3840, 2160, 4, "noise"), (7680, 4320, 4, "grad"), (7680, 4320, 4, "blocks"),
]


def main():
    r = ref()
    assert r.L.ref_supports_sse41() == 1
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(os.path.join(gold, "small"), exist_ok=True)
    kat = []
    for (w, h, c, kind) in CASES:
        img = fpng_amd.synth_image(kind, w, h, c)
        entry = {"w": w, "h": h, "c": c, "kind": kind, "seed": 12345, "flags": {}}
        for fl in (0, 1, 2):
            png = r.encode(img, w, h, c, fl)
            entry["flags"][str(fl)] = {"size": len(png), "sha256": hashlib.sha256(png).hexdigest(),
                                       "btype": (png[60] >> 1) & 3}
            if w * h <= 256:
                with open(os.path.join(gold, "small", f"{kind}_{w}x{h}x{c}_f{fl}.png"), "wb") as f:
                    f.write(png)
        kat.append(entry)
        print(entry["w"], entry["h"], entry["c"], kind, entry["flags"]["0"]["size"])
    with open(os.path.join(gold, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    rng = np.random.default_rng(20240923)
    imgs, meta, outs0, outs1 = [], [], [], []
    for i in range(120):
        img, w, h, c = fuzz_image(rng)
        imgs.append(img.reshape(-1))
        meta.append((w, h, c))
        outs0.append(np.frombuffer(r.encode(img, w, h, c, 0), dtype=np.uint8))
        outs1.append(np.frombuffer(r.encode(img, w, h, c, 1), dtype=np.uint8))
    np.savez_compressed(os.path.join(gold, "fuzz_cases.npz"), meta=np.array(meta, dtype=np.int32),
                        img_off=np.cumsum([0] + [a.size for a in imgs]), img=np.concatenate(imgs),
                        o0_off=np.cumsum([0] + [a.size for a in outs0]), o0=np.concatenate(outs0),
                        o1_off=np.cumsum([0] + [a.size for a in outs1]), o1=np.concatenate(outs1))
    print("wrote", gold)


if __name__ == "__main__":
    main()
