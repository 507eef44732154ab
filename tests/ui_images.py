"""Deterministic screenshot-like test content (TEST INFRASTRUCTURE; golden sizes + sha256 of the reference's files for these
images: tests/golden/ui.json, made by oracle/make_golden_ui.py).  Pure integer numpy: the same pixels everywhere.

  glyphs   rows of 5x7 bitmap "text" in a few colours on a flat background, a 1-px grey fringe at glyph edges (anti-aliasing)
  panels   flat panels with 1-px borders and anti-aliased (two-tone) edges, a few nested, on a flat desktop colour
  dither   horizontal gradients through a 4x4 ordered-dither matrix: short period patterns, no long runs
  matte    RGBA: flat colour regions whose ALPHA is a soft-edged matte -- exact runs of hundreds of pixels that cross the encoder's
           256-pixel super-windows, broken by 3-px ramps

All are uint8 arrays [h, w, c]."""
import numpy as np

_FONT = [  # 5x7 glyphs, one int per row (5 bits), a handful is enough for texture
    [0x0E, 0x11, 0x11, 0x1F, 0x11, 0x11, 0x11], [0x1E, 0x11, 0x1E, 0x11, 0x11, 0x11, 0x1E], [0x0E, 0x11, 0x10, 0x10, 0x10, 0x11, 0x0E],
    [0x1E, 0x11, 0x11, 0x11, 0x11, 0x11, 0x1E], [0x1F, 0x10, 0x1E, 0x10, 0x10, 0x10, 0x1F], [0x11, 0x11, 0x1F, 0x11, 0x11, 0x11, 0x11],
    [0x0E, 0x04, 0x04, 0x04, 0x04, 0x04, 0x0E], [0x11, 0x12, 0x1C, 0x12, 0x11, 0x11, 0x11], [0x11, 0x1B, 0x15, 0x11, 0x11, 0x11, 0x11],
    [0x0E, 0x11, 0x11, 0x11, 0x11, 0x11, 0x0E], [0x1F, 0x04, 0x04, 0x04, 0x04, 0x04, 0x04], [0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00]]


def _lcg(n, seed):
    """n pseudo-random uint32 values (a plain 64-bit LCG in Python integers: identical everywhere)"""
    out = np.empty(n, dtype=np.uint32)
    s = seed
    for i in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
        out[i] = s >> 33
    return out


def glyphs(w, h, c=3, seed=1):
    bg = np.array([250, 250, 246, 255][:c], dtype=np.uint8)
    img = np.empty((h, w, c), dtype=np.uint8)
    img[:] = bg
    cols, rows = (w - 16) // 6, (h - 12) // 12
    r = _lcg(rows * cols + rows, seed)
    inks = np.array([[20, 20, 24, 255], [180, 30, 30, 255], [30, 60, 170, 255], [60, 60, 60, 255]], dtype=np.uint8)[:, :c]
    mask = np.zeros((h, w), dtype=np.uint8)
    ink_of = np.zeros((h, w), dtype=np.uint8)
    for ry in range(rows):
        line_len = 10 + int(r[rows * cols + ry]) % max(1, cols - 10)
        for cx in range(min(cols, line_len)):
            g = _FONT[int(r[ry * cols + cx]) % len(_FONT)]
            x0, y0 = 8 + cx * 6, 6 + ry * 12
            for k in range(7):
                for b in range(5):
                    if (g[k] >> (4 - b)) & 1:
                        mask[y0 + k, x0 + b] = 2
            ink_of[y0:y0 + 7, x0:x0 + 6] = (ry // 3) % len(inks)
    fr = np.zeros_like(mask)  # a 1-px fringe left/right of ink
    fr[:, 1:] |= (mask[:, :-1] == 2)
    fr[:, :-1] |= (mask[:, 1:] == 2)
    fringe = (fr == 1) & (mask == 0)
    for k in range(len(inks)):
        sel = (mask == 2) & (ink_of == k)
        img[sel] = inks[k]
        half = ((inks[k].astype(np.uint16) + bg) // 2).astype(np.uint8)
        img[fringe & (ink_of == k)] = half
    return img


def panels(w, h, c=3, seed=2):
    img = np.empty((h, w, c), dtype=np.uint8)
    img[:] = np.array([58, 110, 165, 255][:c], dtype=np.uint8)
    r = _lcg(64, seed)
    for k in range(12):
        x0, y0 = int(r[4 * k]) % (w * 3 // 4), int(r[4 * k + 1]) % (h * 3 // 4)
        pw, ph = 80 + int(r[4 * k + 2]) % (w // 3), 60 + int(r[4 * k + 3]) % (h // 3)
        x1, y1 = min(w - 1, x0 + pw), min(h - 1, y0 + ph)
        face = np.array([236 - 3 * k, 236 - 2 * k, 240 - k, 255][:c], dtype=np.uint8)
        edge = np.array([96, 96, 104, 255][:c], dtype=np.uint8)
        img[y0:y1, x0:x1] = face
        img[y0:y1, x0] = edge; img[y0:y1, x1 - 1] = edge; img[y0, x0:x1] = edge; img[y1 - 1, x0:x1] = edge
        soft = ((edge.astype(np.uint16) + face) // 2).astype(np.uint8)  # anti-aliased inner line
        if x1 - x0 > 4 and y1 - y0 > 4:
            img[y0 + 1:y1 - 1, x0 + 1] = soft; img[y0 + 1, x0 + 1:x1 - 1] = soft
            img[y0 + 2:y0 + 20, x0 + 2:x1 - 2] = np.array([40, 70, 140, 255][:c], dtype=np.uint8)  # a title bar
    return img


def dither(w, h, c=3, seed=3):
    bayer = np.array([[0, 8, 2, 10], [12, 4, 14, 6], [3, 11, 1, 9], [15, 7, 13, 5]], dtype=np.uint16)
    x = np.arange(w, dtype=np.uint32)[None, :]
    y = np.arange(h, dtype=np.uint32)[:, None]
    t = bayer[(y % 4), (x % 4)]
    img = np.empty((h, w, c), dtype=np.uint8)
    for ch in range(min(c, 3)):
        level = (x * (255 * 16) // max(1, w - 1) + ch * 400 + (y // 64) * 48) % (256 * 16)  # 12-bit ramp, offset per band of rows
        img[:, :, ch] = np.minimum(255, (level + t) // 16).astype(np.uint8) & 0xF8  # quantised to 5 bits: the dither shows
    if c == 4:
        img[:, :, 3] = 255
    return img


def matte(w, h, seed=4):
    img = np.empty((h, w, 4), dtype=np.uint8)
    r = _lcg(h // 40 + 4, seed)
    x = np.arange(w, dtype=np.int64)
    for band in range((h + 39) // 40):
        y0, y1 = band * 40, min(h, band * 40 + 40)
        col = np.array([(37 * band) % 256, (91 * band + 60) % 256, (53 * band + 120) % 256], dtype=np.uint8)
        edge = 200 + int(r[band]) % max(1, w - 400)  # the matte's edge: opaque left of it, a 3-px ramp, transparent right of it
        alpha = np.clip((edge + 3 - x) * 64, 0, 255).astype(np.uint8)
        img[y0:y1, :, :3] = col
        img[y0:y1, :, 3] = alpha[None, :]
    return img


def all_images():
    """name -> (array, w, h, c)"""
    out = {}
    for name, fn in (("glyphs", glyphs), ("panels", panels), ("dither", dither)):
        for (w, h, c) in ((1920, 1080, 3), (3840, 2160, 4)):
            a = fn(w, h, c)
            out[f"{name}_{w}x{h}x{c}"] = (np.ascontiguousarray(a), w, h, c)
    for (w, h) in ((1920, 1080), (3840, 2160)):
        out[f"matte_{w}x{h}x4"] = (np.ascontiguousarray(matte(w, h)), w, h, 4)
    return out
