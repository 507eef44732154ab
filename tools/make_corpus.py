#!/usr/bin/env python
"""A small corpus of ORDINARY PNG files (written with zlib: grey, palette, RGB, RGBA, 16-bit, all row filters) derived from
the natural-image fixture, for the fpng_amd_test harness's corpus workflow (the reference's README numbers come from running
fpng_test over a directory of images):   python tools/make_corpus.py <dir>"""
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))


def write_png(path, arr, ctype, extra=b"", depth=8):
    """arr: uint8 (h, w, samples); rows are stored with filter 0 or 2 (Up) -- zlib does the rest."""
    h, w = arr.shape[:2]
    a = arr.reshape(h, -1).astype(np.int16)
    up = np.vstack([np.zeros((1, a.shape[1]), np.int16), a[:-1]])
    rows = []
    for y in range(h):
        rows.append(bytes([2]) + ((a[y] - up[y]) & 0xFF).astype(np.uint8).tobytes() if y % 3 else bytes([0]) + a[y].astype(np.uint8).tobytes())
    z = zlib.compress(b"".join(rows), 6)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + extra + chunk(b"IDAT", z[: len(z) // 2]) +
                chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


def main(out_dir):
    import real_image
    os.makedirs(out_dir, exist_ok=True)
    # (the drop-in's door imports torch -- a minute or two on a fresh box -- so it is only opened when asked for)
    rgb = real_image.rgb_pixels(__import__("dropin").decode if os.environ.get("FPNG_CORPUS_DROPIN") else __import__("cpu_ref").ref().decode)
    h, w, _ = rgb.shape
    write_png(os.path.join(out_dir, "photo_rgb.png"), rgb, 2)
    write_png(os.path.join(out_dir, "photo_crop_odd.png"), rgb[101:614, 33:550], 2)
    write_png(os.path.join(out_dir, "photo_half.png"), rgb[::2, ::2], 2)
    write_png(os.path.join(out_dir, "photo_wide_strip.png"), np.tile(rgb[300:364], (1, 6, 1)), 2)
    grey = (rgb.astype(np.uint32) @ np.array([77, 150, 29]) >> 8).astype(np.uint8)
    write_png(os.path.join(out_dir, "photo_grey.png"), grey[:, :, None], 0)
    rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)
    rgba[:, :, 3] = np.where(grey > 40, 255, grey * 4)  # translucent shadows
    write_png(os.path.join(out_dir, "photo_rgba.png"), rgba, 6)
    q = (rgb >> 6).astype(np.uint8)  # 64-colour palette version: long exact runs
    idx = (q[:, :, 0] << 4) | (q[:, :, 1] << 2) | q[:, :, 2]
    pal = np.array([[(i >> 4) * 85, ((i >> 2) & 3) * 85, (i & 3) * 85] for i in range(64)], dtype=np.uint8)
    write_png(os.path.join(out_dir, "photo_palette64.png"), idx[:, :, None], 3, extra=chunk(b"PLTE", pal.tobytes()))
    write_png(os.path.join(out_dir, "photo_tiled_2x2.png"), np.tile(rgb, (2, 2, 1)), 2)
    # screenshot-like content (tests/ui_images.py: glyph rows, panels with anti-aliased edges, ordered dither, alpha mattes) at
    # 1080p RGB and 4K RGBA -- what the reference's README corpus has and a photograph does not: long exact runs, few colours
    import ui_images
    for name, (arr, uw, uh, uc) in sorted(ui_images.all_images().items()):
        write_png(os.path.join(out_dir, f"ui_{name}.png"), arr, 2 if uc == 3 else 6)
    with open(os.path.join(out_dir, "fpng_written.png"), "wb") as f:  # a file fpng itself wrote
        f.write(real_image.fixture_bytes())
    print(sorted(os.listdir(out_dir)))


if __name__ == "__main__":
    main(sys.argv[1])
