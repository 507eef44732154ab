"""The fpng_amd_test harness reads arbitrary PNGs with tools/png_loader.h (the role lodepng plays in the reference's harness,
fpng_test.cpp:1116-1122).  CPU test: PNGs of every supported colour type / bit depth, all five row filters, stored / fixed /
dynamic Deflate blocks, several IDAT chunks -- written here with zlib -- must come back as the RGBA8 pixels they encode."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from cpu_ref import ROOT


@pytest.fixture(scope="module")
def loader():
    src = os.path.join(ROOT, "tests", "cpp", "png_loader_shim.cpp")
    hdr = os.path.join(ROOT, "tools", "png_loader.h")
    so = os.path.join(ROOT, "fpng_amd", "lib", "libfpng_test_pngloader.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wall", src, "-o", so])
    L = C.CDLL(so)
    L.shim_load_png.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_char_p, C.c_size_t]

    def load(png):
        b = np.frombuffer(png, dtype=np.uint8)
        out = np.zeros(1 << 24, dtype=np.uint8)
        w, h = C.c_uint32(0), C.c_uint32(0)
        err = C.create_string_buffer(256)
        ok = L.shim_load_png(b.ctypes.data, b.size, out.ctypes.data, out.size, C.byref(w), C.byref(h), err, 256)
        return (out[: w.value * h.value * 4].reshape(h.value, w.value, 4).copy() if ok else None), err.value.decode()
    return load


def _chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))


def _filter_rows(rows, bpp):
    """rows: list of bytes (unfiltered scanlines) -> filtered stream, filter type y % 5."""
    out = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        f = y % 5
        cur = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                p = 0
            elif f == 1:
                p = a
            elif f == 2:
                p = b
            elif f == 3:
                p = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[i] = (v - p) & 0xFF
        out.append(f)
        out += cur
        prev = row
    return bytes(out)


def _png(w, h, ctype, depth, rows, extra=b"", level=6, split=1):
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, chans * depth // 8)
    z = zlib.compress(_filter_rows(rows, bpp), level)
    step = (len(z) + split - 1) // split
    idat = b"".join(_chunk(b"IDAT", z[i:i + step]) for i in range(0, len(z), step))
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + extra + idat + _chunk(b"IEND", b""))


def test_every_colour_type_depth_and_filter(loader):
    rng = np.random.default_rng(3)
    w, h = 37, 23
    # RGB / RGBA 8 bit, smooth content so that the filters matter; compression levels 0 (stored), 1 (fixed-ish), 9 (dynamic)
    for ctype, chans in ((2, 3), (6, 4)):
        img = (np.add.outer(np.arange(h) * 3, np.arange(w) * 5)[:, :, None] + rng.integers(0, 4, (h, w, chans))).astype(np.uint8)
        for level, split in ((0, 1), (1, 3), (9, 2)):
            got, err = loader(_png(w, h, ctype, 8, [bytes(r.tobytes()) for r in img], level=level, split=split))
            assert got is not None, err
            assert np.array_equal(got[:, :, :chans], img) and (chans == 4 or (got[:, :, 3] == 255).all())
    # 16-bit RGBA: the high byte of every sample
    img16 = rng.integers(0, 65536, (h, w, 4)).astype(">u2")
    got, err = loader(_png(w, h, 6, 16, [r.tobytes() for r in img16]))
    assert got is not None and np.array_equal(got, (img16.astype(np.uint16) >> 8).astype(np.uint8)), err
    # grey 8 and grey + alpha
    g = rng.integers(0, 256, (h, w)).astype(np.uint8)
    got, _ = loader(_png(w, h, 0, 8, [r.tobytes() for r in g]))
    assert np.array_equal(got[:, :, 0], g) and np.array_equal(got[:, :, 2], g) and (got[:, :, 3] == 255).all()
    ga = rng.integers(0, 256, (h, w, 2)).astype(np.uint8)
    got, _ = loader(_png(w, h, 4, 8, [r.tobytes() for r in ga]))
    assert np.array_equal(got[:, :, 1], ga[:, :, 0]) and np.array_equal(got[:, :, 3], ga[:, :, 1])
    # grey 1 / 2 / 4 bits (packed, rows padded to bytes), scaled to 0..255
    for depth in (1, 2, 4):
        v = rng.integers(0, 1 << depth, (h, w)).astype(np.uint8)
        rows = []
        for r in v:
            bits = "".join(format(int(x), f"0{depth}b") for x in r)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
        got, err = loader(_png(w, h, 0, depth, rows))
        assert got is not None and np.array_equal(got[:, :, 0], (v.astype(np.uint32) * 255 // ((1 << depth) - 1)).astype(np.uint8)), (depth, err)
    # palette 8 and 4 bit with tRNS
    pal = rng.integers(0, 256, (16, 3)).astype(np.uint8)
    trns = bytes([0, 128, 255])
    idx = rng.integers(0, 16, (h, w)).astype(np.uint8)
    extra = _chunk(b"PLTE", pal.tobytes()) + _chunk(b"tRNS", trns)
    for depth in (8, 4):
        if depth == 8:
            rows = [r.tobytes() for r in idx]
        else:
            rows = [bytes((int(r[i]) << 4) | (int(r[i + 1]) if i + 1 < w else 0) for i in range(0, w, 2)) for r in idx]
        got, err = loader(_png(w, h, 3, depth, rows, extra=extra))
        assert got is not None, err
        assert np.array_equal(got[:, :, :3], pal[idx])
        alpha = np.array([trns[i] if i < len(trns) else 255 for i in range(16)], dtype=np.uint8)
        assert np.array_equal(got[:, :, 3], alpha[idx])


def test_the_reference_example_and_damaged_files(loader):
    """An fpng-written file is an ordinary PNG (one IDAT, an extra fdEC chunk): the general loader and the fpng decoder agree."""
    import real_image
    png = real_image.fixture_bytes()
    got, err = loader(png)
    assert got is not None, err
    g = real_image.gold()
    import hashlib
    assert got.shape == (g["h"], g["w"], 4) and (got[:, :, 3] == 255).all()
    assert hashlib.sha256(np.ascontiguousarray(got[:, :, :3]).tobytes()).hexdigest() == g["pixels_sha256"]
    assert loader(png[:1000])[0] is None and loader(b"not a png at all" * 10)[0] is None
    interlaced = bytearray(_png(4, 4, 2, 8, [bytes(12)] * 4))
    interlaced[28] = 1  # (says Adam7, holds one pass: the sizes cannot match; the IHDR CRC is not looked at by this loader)
    assert loader(bytes(interlaced))[0] is None


def test_corpus_files_of_screenshot_like_content(loader, tmp_path):
    """The corpus workflow's ordinary PNGs of screenshot-like content (tools/make_corpus.py writes them from tests/ui_images.py:
    Up-filtered rows, two IDAT chunks) come back through the harness's loader as the generators' pixels."""
    import importlib.util
    import ui_images
    spec = importlib.util.spec_from_file_location("make_corpus", os.path.join(ROOT, "tools", "make_corpus.py"))
    mc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mc)
    for arr, ctype in ((ui_images.glyphs(1920, 1080, 3), 2), (ui_images.matte(1920, 1080), 6), (ui_images.dither(640, 360, 4), 6)):
        path = str(tmp_path / "ui.png")
        mc.write_png(path, arr, ctype)
        got, err = loader(open(path, "rb").read())
        assert got is not None, err
        c = arr.shape[2]
        assert np.array_equal(got[:, :, :c], arr) and (c == 4 or (got[:, :, 3] == 255).all())


def _png_adam7(w, h, ctype, depth, samples, extra=b""):
    """samples: uint array (h, w, chans) of `depth`-bit values -> an INTERLACED PNG: seven sub-images, each filtered like an image"""
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, chans * depth // 8)
    stream = bytearray()
    for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = []
        for r in sub:
            if depth == 8:
                rows.append(r.astype(np.uint8).tobytes())
            elif depth == 16:
                rows.append(r.astype(">u2").tobytes())
            else:
                bits = "".join(format(int(v), f"0{depth}b") for v in r.reshape(-1))
                bits += "0" * (-len(bits) % 8)
                rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
        stream += _filter_rows(rows, bpp)
    z = zlib.compress(bytes(stream), 6)
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) + extra + _chunk(b"IDAT", z[: len(z) // 3]) +
            _chunk(b"IDAT", z[len(z) // 3:]) + _chunk(b"IEND", b""))


def test_interlaced_files(loader):
    """Adam7 (RFC 2083 section 2.6): the seven passes of every colour type and depth, at sizes where some passes are empty (1 x 1,
    2 x 3, 5 x 1), narrow, and ordinary -- lodepng, the reference harness's loader, reads such files too."""
    rng = np.random.default_rng(7)
    for (w, h) in ((1, 1), (2, 3), (5, 1), (1, 9), (8, 8), (9, 17), (37, 23), (64, 5)):
        for ctype, chans, depths in ((0, 1, (1, 2, 4, 8, 16)), (2, 3, (8, 16)), (3, 1, (1, 2, 4, 8)), (4, 2, (8, 16)), (6, 4, (8, 16))):
            for depth in depths:
                v = rng.integers(0, 1 << depth, (h, w, chans))
                extra = b""
                if ctype == 3:
                    pal = rng.integers(0, 256, (1 << depth, 3)).astype(np.uint8)
                    extra = _chunk(b"PLTE", pal.tobytes())
                got, err = loader(_png_adam7(w, h, ctype, depth, v, extra))
                assert got is not None, (w, h, ctype, depth, err)
                v8 = (v >> 8) if depth == 16 else ((v * 255 // ((1 << depth) - 1)) if ctype == 0 and depth < 8 else v)
                if ctype == 0:
                    exp = np.concatenate([v8.repeat(3, axis=2), np.full((h, w, 1), 255)], axis=2)
                elif ctype == 2:
                    exp = np.concatenate([v8, np.full((h, w, 1), 255)], axis=2)
                elif ctype == 3:
                    exp = np.concatenate([pal[v[:, :, 0]], np.full((h, w, 1), 255)], axis=2)
                elif ctype == 4:
                    exp = np.concatenate([v8[:, :, :1].repeat(3, axis=2), v8[:, :, 1:]], axis=2)
                else:
                    exp = v8
                assert np.array_equal(got, exp.astype(np.uint8)), (w, h, ctype, depth)
