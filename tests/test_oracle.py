"""CPU tests: the oracle (oracle/fpng_oracle.c) against the committed golden vectors and, when the
reference build is present (dev container / prebuilt oracle/_ref), against the reference itself."""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

from cpu_ref import ROOT, fuzz_image, have_ref, oracle, ref

GOLD = os.path.join(ROOT, "tests", "golden")


def _kat():
    with open(os.path.join(GOLD, "kat.json")) as f:
        return json.load(f)


def _synth(kind, w, h, c):
    import fpng_amd
    return fpng_amd.synth_image(kind, w, h, c)


@pytest.mark.parametrize("entry", [e for e in _kat() if e["w"] * e["h"] <= 1920 * 1080],
                         ids=lambda e: f'{e["kind"]}_{e["w"]}x{e["h"]}x{e["c"]}')
def test_oracle_matches_golden_kat(entry, built_lib):
    img = _synth(entry["kind"], entry["w"], entry["h"], entry["c"])
    for fl, exp in entry["flags"].items():
        png = oracle().encode(img, entry["w"], entry["h"], entry["c"], int(fl))
        assert len(png) == exp["size"]
        assert hashlib.sha256(png).hexdigest() == exp["sha256"]


def test_oracle_matches_golden_small_files(built_lib):
    n = 0
    for name in sorted(os.listdir(os.path.join(GOLD, "small"))):
        kind, dims, fl = name[:-4].split("_")
        w, h, c = (int(v) for v in dims.split("x"))
        with open(os.path.join(GOLD, "small", name), "rb") as f:
            exp = f.read()
        assert oracle().encode(_synth(kind, w, h, c), w, h, c, int(fl[1:])) == exp, name
        n += 1
    assert n >= 20


def test_oracle_matches_golden_fuzz():
    z = np.load(os.path.join(GOLD, "fuzz_cases.npz"))
    for i, (w, h, c) in enumerate(z["meta"]):
        img = z["img"][z["img_off"][i]:z["img_off"][i + 1]]
        for fl, key in ((0, "o0"), (1, "o1")):
            exp = z[key][z[key + "_off"][i]:z[key + "_off"][i + 1]].tobytes()
            assert oracle().encode(img, int(w), int(h), int(c), fl) == exp, (i, w, h, c, fl)


def test_oracle_output_is_valid_png_zlib():
    """Independent check: zlib inflates the IDAT payload back to the filtered rows."""
    rng = np.random.default_rng(3)
    for _ in range(40):
        img, w, h, c = fuzz_image(rng)
        for fl in (0, 1, 2):
            png = oracle().encode(img, w, h, c, fl)
            idat_len = int.from_bytes(png[50:54], "big")
            assert png[54:58] == b"IDAT" and len(png) == 58 + idat_len + 16
            raw = zlib.decompress(png[58:58 + idat_len])
            assert len(raw) == (w * c + 1) * h
            rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, w * c + 1)
            src = img.reshape(h, w * c).astype(np.int32)
            stored = (png[60] >> 1) & 3 == 0
            for y in range(h):
                if y == 0 or stored:
                    assert rows[y, 0] == 0 and (rows[y, 1:] == src[y]).all()
                else:
                    assert rows[y, 0] == 2 and (rows[y, 1:] == ((src[y] - src[y - 1]) & 0xFF)).all()
            assert zlib.crc32(png[54:58 + idat_len]) == int.from_bytes(png[58 + idat_len:62 + idat_len], "big")


def test_oracle_checksums_match_zlib():
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 8, 9, 5551, 5552, 5553, 100000):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        assert oracle().crc32(d) == zlib.crc32(d.tobytes())
        assert oracle().adler32(d) == zlib.adler32(d.tobytes())
        k = n // 3
        assert oracle().crc32(d[k:], oracle().crc32(d[:k])) == zlib.crc32(d.tobytes())
        assert oracle().adler32(d[k:], oracle().adler32(d[:k])) == zlib.adler32(d.tobytes())


def test_oracle_bad_arguments():
    img = np.zeros(64, dtype=np.uint8)
    assert oracle().encode(img, 0, 1, 3) is None       # reference fpng.cpp:1670
    assert oracle().encode(img, 1, 0, 3) is None
    assert oracle().encode(img, 1, 1, 2) is None       # reference fpng.cpp:1676
    assert oracle().encode(img, 1, 1, 5) is None
    assert oracle().encode(img, (1 << 24) + 1, 1, 3) is None


@pytest.mark.skipif(not have_ref(), reason="reference build not available")
def test_oracle_vs_reference_fuzz():
    """The pin: byte-for-byte against the unmodified reference on the edge-case recipe."""
    rng = np.random.default_rng(11)
    raw = 0
    for i in range(1500):
        img, w, h, c = fuzz_image(rng)
        for fl in (0, 1, 2):
            a, b = oracle().encode(img, w, h, c, fl), ref().encode(img, w, h, c, fl)
            assert a == b, (i, w, h, c, fl)
        raw += (b[60] >> 1) & 3 == 0
    assert raw > 100  # the recipe must actually reach the stored-block fallback


@pytest.mark.skipif(not have_ref(), reason="reference build not available")
def test_oracle_vs_reference_adjust_freq_quirk():
    """Large skewed histograms: vertical deltas in {0,+1,-1} plus singleton outliers."""
    rng = np.random.default_rng(12)
    w, h, c = 1024, 256, 3
    d = rng.choice(np.array([0, 1, 255], dtype=np.uint8), size=(h, w * c), p=[0.9, 0.05, 0.05])
    out = rng.choice(np.arange(3, 250), size=200, replace=False)
    pos = rng.choice(h * w * c, size=200, replace=False)
    d.reshape(-1)[pos] = out.astype(np.uint8)
    img = np.cumsum(d.astype(np.int64), axis=0).astype(np.uint8)
    for fl in (0, 1):
        assert oracle().encode(img, w, h, c, fl) == ref().encode(img, w, h, c, fl)


@pytest.mark.skipif(not have_ref(), reason="reference build not available")
def test_reference_decodes_oracle_output():
    rng = np.random.default_rng(13)
    for _ in range(30):
        img, w, h, c = fuzz_image(rng)
        for fl in (0, 1):
            st, out, ww, hh, cc = ref().decode(oracle().encode(img, w, h, c, fl), c)
            assert st == 0 and (ww, hh, cc) == (w, h, c)
            assert (out == img.reshape(-1)).all()


def test_oracle_band_concatenation_equals_whole():
    """Row independence: band bit strings concatenate to the whole image's stream."""
    rng = np.random.default_rng(17)
    for _ in range(25):
        img, w, h, c = fuzz_image(rng, force_dims=(int(rng.integers(1, 80)), int(rng.integers(2, 12))))
        cuts = sorted(set([0, h] + [int(v) for v in rng.integers(1, h, 2)]))
        bits_total, ints = 0, 0
        for y0, y1 in zip(cuts[:-1], cuts[1:]):
            bits, buf, s1, s2, ln = oracle().band_1pass(img, w, h, c, y0, y1)
            ints |= int.from_bytes(buf.tobytes(), "little") << bits_total
            bits_total += bits
        wbits, wbuf, _, _, _ = oracle().band_1pass(img, w, h, c, 0, h)
        assert bits_total == wbits and ints == int.from_bytes(wbuf.tobytes(), "little")


def test_oracle_on_the_natural_image():
    """The restatement against the reference's bytes on a photograph (tests/golden/real.json, made by
    oracle/make_golden_real.py from the reference's example.png): every variant, flags 0/1/2."""
    import hashlib
    import real_image
    r = ref() if have_ref() else None
    dec = r.decode if r else __import__("dropin").decode
    imgs = real_image.variants(real_image.rgb_pixels(dec))
    g = real_image.gold()["variants"]
    for k, img in imgs.items():
        h, w, c = img.shape
        for fl in (0, 1, 2):
            png = oracle().encode(img, w, h, c, fl)
            assert len(png) == g[k]["flags"][str(fl)]["size"]
            assert hashlib.sha256(png).hexdigest() == g[k]["flags"][str(fl)]["sha256"], (k, fl)


@pytest.mark.skipif(not have_ref(), reason="reference build not available")
def test_oracle_vs_reference_wide_rows():
    """The checker itself on the wide-row generator the -m gpu suite uses (tools/gpu_wide_fuzz.py): rows of 257..4100 pixels,
    runs crossing every 256-pixel border -- identical to the unmodified reference for all three flag values."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_wide_fuzz", os.path.join(ROOT, "tools", "gpu_wide_fuzz.py"))
    src = open(spec.origin).read().split("def main():")[0].replace("import numpy as np, torch, fpng_amd", "import numpy as np")
    ns = {"__file__": spec.origin}
    exec(compile(src, spec.origin, "exec"), ns)
    rng = np.random.default_rng(77)
    for _ in range(300):
        img, w, h, c = ns["wide_image"](rng)
        for fl in (0, 1, 2):
            assert oracle().encode(img, w, h, c, fl) == ref().encode(img, w, h, c, fl), (w, h, c, fl)


def _is_stored(png):
    return (png[58 + 2] & 6) == 0  # (58 = signature + IHDR + fdEC + the IDAT's prefix; the block's type bits sit behind the zlib header)


@pytest.mark.skipif(not have_ref(), reason="reference build not available")
def test_oracle_vs_reference_at_the_stored_or_compressed_boundary():
    """The reference keeps the compressed stream unless its coder "ran out of buffer" -- a rule with a closed form in the final bit
    position and 8 bytes of slack (reference src/fpng.cpp:567-588 / :1728-1758).  Images whose first K pixels are noise and
    whose rest is flat: the compressed size grows with K; around the K where the outcome flips (and for some K on either side)
    the checker must write the reference's bytes, 1-pass and 2-pass."""
    rng = np.random.default_rng(2718)
    flips = 0
    for (w, h, c) in ((64, 32, 4), (61, 17, 3), (256, 9, 4), (85, 30, 3), (33, 33, 4), (1024, 3, 3), (7, 200, 4)):
        noise = rng.integers(0, 256, (w * h, c), dtype=np.uint8)
        flat = np.full((w * h, c), 77, dtype=np.uint8)
        for flags in (0, 1):
            def make(k):
                img = flat.copy()
                img[:k] = noise[:k]
                return img.reshape(-1)
            lo, hi = 0, w * h  # invariant: make(lo) compressed, make(hi) stored (noise all over does not compress)
            assert not _is_stored(ref().encode(make(lo), w, h, c, flags))
            if not _is_stored(ref().encode(make(hi), w, h, c, flags)):
                continue  # (a 2-pass table can hold noise in 8 bits a byte: narrow images stay compressed)
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if _is_stored(ref().encode(make(mid), w, h, c, flags)):
                    hi = mid
                else:
                    lo = mid
            outcomes = set()
            for k in range(max(0, hi - 40), min(w * h, hi + 40) + 1):
                img = make(k)
                want = ref().encode(img, w, h, c, flags)
                got = oracle().encode(img, w, h, c, flags)
                assert got == want, (w, h, c, flags, k, len(got), len(want))
                outcomes.add(_is_stored(want))
            flips += outcomes == {False, True}
    assert flips >= 10, flips


def skewed_images(rng, max_bytes=400_000):
    """Images whose FILTERED bytes hold `levels` byte values with counts ratio^0, ratio^1, ...: the optimal prefix code of such a
    histogram is deeper than the 12 bits fpng allows from 13 levels on, so the 2-pass builder's length limiting and, for the large
    ones, adjust_freq32's scaling (reference src/fpng.cpp:909-988, :990-1161) decide the table."""
    out = []
    for levels in (2, 3, 8, 12, 13, 14, 16, 19, 23):
        for c in (3, 4):
            for ratio in (1.3, 1.62, 2.0, 3.0):
                counts = [max(1, int(ratio ** k)) for k in range(levels)]
                if sum(counts) > max_bytes:
                    continue
                vals = rng.permutation(256)[:levels]
                data = np.concatenate([np.full(n, v, np.uint8) for v, n in zip(vals, counts)])
                if rng.random() < 0.5:
                    rng.shuffle(data)
                n = len(data)
                w = max(1, int(np.sqrt(n / c)))
                h = max(1, n // (w * c))
                if w * h * c > n:
                    data = np.concatenate([data, np.full(w * h * c - n, vals[0], np.uint8)])
                f = data[: w * h * c].reshape(h, w * c)
                out.append((np.cumsum(f.astype(np.uint32), axis=0).astype(np.uint8).reshape(-1), w, h, c))  # (Up-filtered: back to f)
    return out


@pytest.mark.skipif(not have_ref(), reason="reference build not available")
def test_oracle_vs_reference_on_skewed_histograms():
    rng = np.random.default_rng(1618)
    imgs = skewed_images(rng)
    assert len(imgs) >= 40
    for img, w, h, c in imgs:
        for flags in (0, 1):
            assert oracle().encode(img, w, h, c, flags) == ref().encode(img, w, h, c, flags), (w, h, c, flags)
