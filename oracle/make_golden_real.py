"""Golden vectors from a NATURAL image (TEST INFRASTRUCTURE): the one real picture the reference ships,
/root/reference/example.png (687x1012 RGB, itself an fpng file), run through the UNMODIFIED reference
(oracle/_ref/libfpng_ref.so).

Run in the dev container (where /root/reference exists):   python oracle/make_golden_real.py

Writes
  tests/golden/real/example_rgb_f1.png   the reference's FPNG_ENCODE_SLOWER encoding of the decoded pixels: the pixel
                                         fixture (the GPU box has no /root/reference) and a known answer in itself
  tests/golden/real.json                 per variant and flags 0/1/2: size + sha256 of the reference's output

Variants (what the reference harness feeds its encoder, fpng_test.cpp:1116-1190):
  rgb        the 24 bpp pixels                                   (source_chans == 3 path)
  rgba       32 bpp, alpha 255                                   (lodepng's LCT_RGBA buffer, fpng_test.cpp:1117)
  rgba_ga    32 bpp, alpha = green (-a, fpng_test.cpp:1147-1152)
  rgb_t4 / rgba_ga_t4   the same pixels tiled 4 x 4 (2748x4048): rows cross many 256-pixel super-windows, the copies
             make long exact vertical repeats only at tile seams -- a natural tier mix at a size that fills the GPU
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SRC = "/root/reference/example.png"


def variants(rgb):
    """name -> uint8 [h, w, c]; rgb = uint8 [h, w, 3].  tests/test_gpu_real_image.py builds the same arrays."""
    h, w, _ = rgb.shape
    rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)
    ga = rgba.copy()
    ga[:, :, 3] = ga[:, :, 1]
    return {
        "rgb": rgb,
        "rgba": rgba,
        "rgba_ga": ga,
        "rgb_t4": np.tile(rgb, (4, 4, 1)),
        "rgba_ga_t4": np.tile(ga, (4, 4, 1)),
    }


def main():
    from cpu_ref import ref
    r = ref()
    src = open(SRC, "rb").read()
    st, px, w, h, c = r.decode(src, 3)
    assert st == 0 and c == 3, (st, c)
    rgb = px.reshape(h, w, 3)
    out_dir = os.path.join(ROOT, "tests", "golden", "real")
    os.makedirs(out_dir, exist_ok=True)
    gold = {"source": "reference example.png decoded by the reference's fpng_decode_memory", "w": w, "h": h,
            "pixels_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(), "variants": {}}
    for name, img in variants(rgb).items():
        img = np.ascontiguousarray(img)
        hh, ww, cc = img.shape
        e = {"w": ww, "h": hh, "c": cc, "pixels_sha256": hashlib.sha256(img.tobytes()).hexdigest(), "flags": {}}
        for fl in (0, 1, 2):
            png = r.encode(img, ww, hh, cc, fl)
            e["flags"][str(fl)] = {"size": len(png), "sha256": hashlib.sha256(png).hexdigest()}
            if name == "rgb" and fl == 1:
                with open(os.path.join(out_dir, "example_rgb_f1.png"), "wb") as f:
                    f.write(png)
                print("example.png itself is the reference's", "2-pass" if png == src else "(not 2-pass)", "encoding")
        gold["variants"][name] = e
        print(name, ww, hh, cc, {k: v["size"] for k, v in e["flags"].items()})
    with open(os.path.join(ROOT, "tests", "golden", "real.json"), "w") as f:
        json.dump(gold, f, indent=1)


if __name__ == "__main__":
    main()
