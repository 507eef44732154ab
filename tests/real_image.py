"""The natural-image fixture (tests/golden/real/, made by oracle/make_golden_real.py from the reference's example.png):
pixels out of the committed fpng file, and the variants the golden hashes were taken on."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold():
    with open(os.path.join(GOLD, "real.json")) as f:
        return json.load(f)


def fixture_bytes():
    with open(os.path.join(GOLD, "real", "example_rgb_f1.png"), "rb") as f:
        return f.read()


def rgb_pixels(decode):
    """decode(png_bytes, desired_chans) -> (status, uint8 array, w, h, c): the reference's decoder when present, else the
    drop-in's CPU decoder.  The pixels are pinned by their sha256 either way."""
    g = gold()
    st, px, w, h, c = decode(fixture_bytes(), 3)
    assert st == 0 and (w, h, c) == (g["w"], g["h"], 3)
    rgb = np.ascontiguousarray(px[: w * h * 3]).reshape(h, w, 3)
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == g["pixels_sha256"]
    return rgb


def variants(rgb):
    h, w, _ = rgb.shape
    rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)
    ga = rgba.copy()
    ga[:, :, 3] = ga[:, :, 1]
    out = {"rgb": rgb, "rgba": rgba, "rgba_ga": ga, "rgb_t4": np.tile(rgb, (4, 4, 1)), "rgba_ga_t4": np.tile(ga, (4, 4, 1))}
    g = gold()["variants"]
    for k, v in out.items():
        out[k] = v = np.ascontiguousarray(v)
        assert hashlib.sha256(v.tobytes()).hexdigest() == g[k]["pixels_sha256"], k
    return out
