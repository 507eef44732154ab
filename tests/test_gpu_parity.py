"""GPU parity tests (-m gpu): the HIP encoder, called through the C ABI, against the oracle, the
committed golden vectors and size-independent properties.  Bit-exact is the bar."""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

from cpu_ref import ROOT, fuzz_image, have_ref, oracle, ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def enc(built_lib):
    import torch
    import fpng_amd
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    e = fpng_amd.Encoder(device=0)
    yield e
    e.close()


def _gpu_encode(enc, imgs, flags=0):
    import torch
    ts = [torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs]
    pngs, modes = enc.encode_tensors(ts, flags)
    return pngs, modes


def _first_diff(a, b):
    n = min(len(a), len(b))
    x = np.frombuffer(a[:n], dtype=np.uint8) != np.frombuffer(b[:n], dtype=np.uint8)
    return int(np.argmax(x)) if x.any() else n


def _assert_same(png, exp, what):
    if png != exp:
        d = _first_diff(png, exp)
        raise AssertionError(f"{what}: sizes {len(png)} vs {len(exp)}, first difference at byte {d}: "
                             f"{png[d:d+8].hex()} vs {exp[d:d+8].hex()}")


def _kat():
    with open(os.path.join(GOLD, "kat.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("entry", _kat(), ids=lambda e: f'{e["kind"]}_{e["w"]}x{e["h"]}x{e["c"]}')
@pytest.mark.parametrize("flags", [0, 1, 2])
def test_golden_kat(enc, entry, flags):
    """Known answers produced by the unmodified reference (tests/golden/kat.json), up to 8K RGBA."""
    import fpng_amd
    w, h, c = entry["w"], entry["h"], entry["c"]
    img = fpng_amd.synth_image(entry["kind"], w, h, c)
    (png,), (mode,) = _gpu_encode(enc, [img], flags)
    exp = entry["flags"][str(flags)]
    assert len(png) == exp["size"]
    assert hashlib.sha256(png).hexdigest() == exp["sha256"]
    assert mode == (1 if exp["btype"] == 0 else 0)


def test_golden_small_files(enc):
    import fpng_amd
    for name in sorted(os.listdir(os.path.join(GOLD, "small"))):
        kind, dims, fl = name[:-4].split("_")
        w, h, c = (int(v) for v in dims.split("x"))
        with open(os.path.join(GOLD, "small", name), "rb") as f:
            exp = f.read()
        (png,), _ = _gpu_encode(enc, [fpng_amd.synth_image(kind, w, h, c)], int(fl[1:]))
        _assert_same(png, exp, name)


def test_golden_fuzz_cases(enc):
    z = np.load(os.path.join(GOLD, "fuzz_cases.npz"))
    imgs, exps = [], []
    for i, (w, h, c) in enumerate(z["meta"]):
        imgs.append(z["img"][z["img_off"][i]:z["img_off"][i + 1]].reshape(int(h), int(w), int(c)))
        exps.append(z["o0"][z["o0_off"][i]:z["o0_off"][i + 1]].tobytes())
    pngs, _ = _gpu_encode(enc, imgs, 0)
    for i, (p, e) in enumerate(zip(pngs, exps)):
        _assert_same(p, e, f"fuzz case {i} {tuple(z['meta'][i])}")
    pngs, _ = _gpu_encode(enc, imgs, 1)
    for i, p in enumerate(pngs):
        _assert_same(p, z["o1"][z["o1_off"][i]:z["o1_off"][i + 1]].tobytes(), f"2-pass fuzz case {i} {tuple(z['meta'][i])}")


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_fuzz_vs_oracle_batched(enc, flags):
    """2000 edge-case images (SURVEY B.3 recipe), encoded as batches in single submissions."""
    rng = np.random.default_rng(100 + flags)
    stored = 0
    for _ in range(10):
        cases = [fuzz_image(rng) for _ in range(200)]
        pngs, modes = _gpu_encode(enc, [c[0] for c in cases], flags)
        for (img, w, h, c), p in zip(cases, pngs):
            _assert_same(p, oracle().encode(img, w, h, c, flags), f"fuzz {w}x{h}x{c}")
        stored += sum(modes)
    assert stored > 100


@pytest.mark.parametrize("c", [3, 4])
def test_widths_around_window_and_chunk_limits(enc, c):
    """Window (64 px) and chunk-cap (63 / 85 px) boundaries, runs crossing windows."""
    rng = np.random.default_rng(7 + c)
    imgs, dims = [], []
    for w in [1, 2, 3, 62, 63, 64, 65, 66, 84, 85, 86, 87, 126, 127, 128, 129, 130, 170, 171, 172, 191, 192, 193, 255, 256,
              257, 500, 1000, 4099]:
        for h in (1, 2, 5):
            for kind in range(4):
                if kind == 0:      # solid
                    img = np.full((h, w, c), 77, dtype=np.uint8)
                elif kind == 1:    # long runs with random breaks
                    img = np.repeat(rng.integers(0, 256, (h, (w + 39) // 40, c), dtype=np.uint8), 40, axis=1)[:, :w]
                elif kind == 2:    # runs of random length 1..200 on the FILTERED rows: make rows identical so Up gives zeros after row 0
                    row = np.repeat(rng.integers(0, 256, (1, w, c), dtype=np.uint8), h, axis=0)
                    brk = rng.random((1, w)) < 0.03
                    row[:, ~brk[0]] = 0
                    row = np.maximum.accumulate(row, axis=1)
                    img = row
                else:              # two-pixel alternation (1-pixel matches -> literal-vs-match rule)
                    a = rng.integers(0, 256, (h, 1, c), dtype=np.uint8)
                    img = np.repeat(a, w, axis=1).copy()
                    img[:, 2::3] = rng.integers(0, 256, (h, len(range(2, w, 3)), c), dtype=np.uint8)
                imgs.append(np.ascontiguousarray(img))
                dims.append((w, h, c))
    for fl in (0, 1):
        pngs, _ = _gpu_encode(enc, imgs, fl)
        for img, (w, h, c_), p in zip(imgs, dims, pngs):
            _assert_same(p, oracle().encode(img, w, h, c_, fl), f"{w}x{h}x{c_} flags={fl}")


def _tier_pattern(rng, kind, w, c):
    """One FILTERED row exercising a particular mix of the row walker's super-window (256 px) tiers."""
    row = rng.integers(0, 256, (w, c), dtype=np.uint8)
    x = np.arange(w)
    if kind == "sparse":           # isolated repeats, some of them straddling 4-pixel lane groups and super-window seams
        rep = rng.random(w) < 0.01
        rep[[i for i in (1, 3, 4, 255, 256, 257, 511, 512) if i < w]] = True
        rep[0] = False
        for i in np.flatnonzero(rep):
            row[i] = row[i - 1]
    elif kind == "lit_then_runs":  # literal super-windows, then run-heavy ones (walker hands the row over mid-way)
        row[x >= (w // 2)] = row[w // 2]
    elif kind == "runs_then_lit":
        row[x < (w // 2)] = 9
    elif kind == "alternate":      # general and literal super-windows alternate: the hand-over streak must reset
        for s in range(0, w, 512):
            row[s:s + 200] = row[s]
    elif kind == "pairs":          # every second pixel repeats: 1-pixel matches everywhere (literal-vs-match rule)
        row[1::2] = row[0:w - (w % 2):2][: len(row[1::2])]
    elif kind == "long_run_mid":   # a run longer than the chunk cap crossing several super-windows
        a, b = w // 5, w - w // 7
        row[a:b] = row[a]
    return row


@pytest.mark.parametrize("c", [3, 4])
def test_super_window_tiers(enc, c):
    """Widths around the 256-pixel super-window and its 4-pixel lane groups, with content that steers each tier."""
    rng = np.random.default_rng(40 + c)
    imgs, dims = [], []
    for w in [252, 256, 257, 259, 260, 511, 512, 513, 516, 768, 1023, 1024, 1025, 1300, 2065]:
        for kind in ("sparse", "lit_then_runs", "runs_then_lit", "alternate", "pairs", "long_run_mid"):
            r0 = _tier_pattern(rng, kind, w, c)
            r1 = (r0.astype(np.uint16) + _tier_pattern(rng, kind, w, c)).astype(np.uint8)   # Up-filtered row 1 = second pattern
            r2 = (r1.astype(np.uint16) + _tier_pattern(rng, "sparse", w, c)).astype(np.uint8)
            imgs.append(np.ascontiguousarray(np.stack([r0, r1, r2])))
            dims.append((w, 3, c))
    for fl in (0, 1):
        pngs, _ = _gpu_encode(enc, imgs, fl)
        for img, (w, h, c_), p in zip(imgs, dims, pngs):
            _assert_same(p, oracle().encode(img, w, h, c_, fl), f"tiers {w}x{h}x{c_} flags={fl}")


@pytest.mark.parametrize("c", [3, 4])
def test_rows_ending_on_a_super_window(enc, c):
    """Rows of at least 256 pixels end inside the 4-pixels-per-lane phase of the walk: the file, and the size of
    the row's final flush unit (it feeds the reference's stored-or-compressed rule, reference fpng.cpp:567-588), for every
    kind of row end -- literal, isolated repeat, run into the last pixel, run through the whole last super-window."""
    import torch
    from fpng_amd import sharded
    from band_backend import OracleBandBackend
    rng = np.random.default_rng(90 + c)
    be = sharded.GpuBandBackend(enc)
    for w in (256, 257, 259, 300, 511, 512, 700, 768, 1000, 1024, 1920, 3840, 4099):   # the last super-window complete or partial
        for end in ("literal", "repeat1", "repeat2", "run_cap", "solid_tail", "sparse"):
            rows = rng.integers(0, 256, (3, w, c), dtype=np.uint8)
            for r in rows:  # the FILTERED rows get the pattern: build them, then integrate over y
                if end == "repeat1":
                    r[w - 1] = r[w - 2]
                elif end == "repeat2":
                    r[w - 2:] = r[w - 3]
                elif end == "run_cap":
                    r[w - 64:] = r[w - 65]
                elif end == "solid_tail":
                    r[w - 256:] = 5
                elif end == "sparse":
                    for i in (3, 64, 200, w - 5, w - 1):
                        r[i] = r[i - 1]
            img = np.ascontiguousarray(np.cumsum(rows.astype(np.uint16), axis=0).astype(np.uint8))
            for fl in (0, 1):
                pngs, _ = _gpu_encode(enc, [img], fl)
                _assert_same(pngs[0], oracle().encode(img, w, 3, c, fl), f"row end {end} {w}x3x{c} flags={fl}")
            # one band = the whole image: its counts straight from the kernels
            t = torch.from_numpy(img).cuda()
            got = be.encode(t, None, w, c, 0, 3, 3, 0, None)
            want = OracleBandBackend(img).encode(None, None, w, c, 0, 3, 3, 0, None)
            assert (got.token_bits, got.last_unit_bits, got.s1, got.s2) == (want.token_bits, want.last_unit_bits, want.s1, want.s2), (end, w, c, got, want)


@pytest.mark.parametrize("c", [3, 4])
def test_maximum_width_rows(enc, c):
    """w = 2^24 (the reference's limit, fpng.cpp:1670): 48 / 64 MiB per row, two rows so that the Up filter runs.
    Mixed content: noisy stretches, long runs (thousands of chunks per row), a few isolated repeats."""
    w, h = 1 << 24, 2
    rng = np.random.default_rng(1000 + c)
    row = rng.integers(0, 256, (w, c), dtype=np.uint8)
    row[w // 8: w // 2] = row[w // 8]                      # one run of 6.3 M pixels
    row[5_000_000:5_000_400:2] = row[5_000_001:5_000_401:2]  # isolated pairs
    img = np.stack([row, np.roll(row, 12345, axis=0)])
    (png,), (mode,) = _gpu_encode(enc, [np.ascontiguousarray(img)], 0)
    _assert_same(png, oracle().encode(img, w, h, c, 0), f"max width {w}x{h}x{c}")


def test_maximum_height(enc):
    """h = 2^24 rows of one pixel (the reference's limit): 16.7 M rows through the row scan and the row search."""
    w, h, c = 1, 1 << 24, 4
    rng = np.random.default_rng(4321)
    img = np.repeat(rng.integers(0, 256, (h // 64, w, c), dtype=np.uint8), 64, axis=0)  # 63 zero rows after each change
    img[1::1000] = rng.integers(0, 256, (len(range(1, h, 1000)), w, c), dtype=np.uint8)
    img = np.ascontiguousarray(img)
    for fl in (0, 1):
        (png,), _ = _gpu_encode(enc, [img], fl)
        _assert_same(png, oracle().encode(img, w, h, c, fl), f"max height flags={fl}")


@pytest.mark.parametrize("c", [3, 4])
def test_longest_possible_local_streams(enc, c):
    """Every filtered byte carries the LONGEST literal code of the 1-pass table and no pixel repeats: each row's local
    stream is as long as the scratch stride allows.  Such an image falls back to stored blocks itself; the images
    around it in the batch (whose local streams are its neighbours in scratch) must come out untouched."""
    lens = np.array(oracle().table_1pass(c)[0][:256])
    worst = np.flatnonzero(lens == lens.max())
    assert len(worst) >= 2
    rng = np.random.default_rng(5 + c)
    w, h = 1300, 37
    filt = worst[rng.integers(0, len(worst), (h, w, c))].astype(np.uint8)
    # no two horizontally adjacent filtered pixels equal (that would start a run): patch channel 0 where needed
    for _ in range(4):
        same = np.all(filt[:, 1:] == filt[:, :-1], axis=2)
        ys, xs = np.nonzero(same)
        filt[ys, xs + 1, 0] = worst[(np.searchsorted(worst, filt[ys, xs + 1, 0]) + 1) % len(worst)]
    assert not np.all(filt[:, 1:] == filt[:, :-1], axis=2).any()
    img = np.cumsum(filt.astype(np.uint32), axis=0).astype(np.uint8)   # Up filter of img gives filt back
    import fpng_amd
    others = [fpng_amd.synth_image(k, 1300, 29, c, seed=70 + i) for i, k in enumerate(("grad", "blocks", "grad"))]
    batch = [others[0], np.ascontiguousarray(img), others[1], np.ascontiguousarray(img[:, ::-1]), others[2]]
    for fl in (0, 1):
        pngs, modes = _gpu_encode(enc, batch, fl)
        for im, p in zip(batch, pngs):
            hh, ww, cc = im.shape
            _assert_same(p, oracle().encode(im, ww, hh, cc, fl), f"worst-case neighbours {ww}x{hh}x{cc} flags={fl}")
        if fl == 0:
            assert modes[1] == 1   # the all-longest-codes image itself is stored


def test_many_tiny_rows_through_assemble(enc):
    """Rows of a few bytes: more than 64 rows under one 1 KiB chunk of assemble_kernel (its slow row walk), deep
    64-ary row searches (h up to 200 000), row seams in nearly every dword."""
    rng = np.random.default_rng(77)
    imgs, dims = [], []
    for (w, h, c) in [(1, 50000, 3), (1, 50000, 4), (2, 70000, 4), (3, 30011, 3), (5, 20000, 4), (1, 200000, 4), (17, 9001, 3)]:
        kind = rng.integers(0, 3)
        if kind == 0:
            img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        elif kind == 1:  # identical rows: Up filter gives zero rows (1-pixel-wide runs of zeros)
            img = np.repeat(rng.integers(0, 256, (1, w, c), dtype=np.uint8), h, axis=0)
        else:            # piecewise constant columns with random breaks
            img = np.repeat(rng.integers(0, 256, ((h + 99) // 100, w, c), dtype=np.uint8), 100, axis=0)[:h]
        imgs.append(np.ascontiguousarray(img))
        dims.append((w, h, c))
    for fl in (0, 1):
        pngs, _ = _gpu_encode(enc, imgs, fl)
        for img, (w, h, c), p in zip(imgs, dims, pngs):
            _assert_same(p, oracle().encode(img, w, h, c, fl), f"tiny rows {w}x{h}x{c} flags={fl}")


def test_mixed_batch_shapes_and_channels(enc):
    import fpng_amd
    imgs = [fpng_amd.synth_image(k, w, h, c) for (k, w, h, c) in
            [("grad", 640, 480, 3), ("blocks", 333, 77, 4), ("noise", 50, 50, 4), ("solid", 1, 1, 3), ("grad", 1921, 3, 4),
             ("noise", 7, 300, 3), ("blocks", 1024, 1024, 3), ("grad", 2, 2, 4)]]
    pngs, _ = _gpu_encode(enc, imgs, 0)
    assert enc.phase_names()[0] == "encode_rows"   # the default whole-image pipeline (one walk into scratch streams + assemble)
    for img, p in zip(imgs, pngs):
        h, w, c = img.shape
        _assert_same(p, oracle().encode(img, w, h, c, 0), f"{w}x{h}x{c}")


def test_full_size_properties_8k(enc):
    """BASELINE north-star size: checks that do not need a CPU encode of the full image.
    zlib inflates the payload to exactly the Up-filtered image; CRC and Adler verify."""
    import fpng_amd
    w, h, c = 7680, 4320, 4
    img = fpng_amd.synth_image("grad", w, h, c, seed=4242)
    (png,), (mode,) = _gpu_encode(enc, [img], 0)
    assert mode == 0 and png[:8] == b"\x89PNG\r\n\x1a\n"
    idat = int.from_bytes(png[50:54], "big")
    assert len(png) == 58 + idat + 16
    raw = np.frombuffer(zlib.decompress(png[58:58 + idat]), dtype=np.uint8).reshape(h, w * c + 1)
    src = img.reshape(h, w * c)
    assert (raw[0, 1:] == src[0]).all() and raw[0, 0] == 0 and (raw[1:, 0] == 2).all()
    assert (raw[1:, 1:] == (src[1:] - src[:-1])).all()
    assert zlib.crc32(png[54:58 + idat]) == int.from_bytes(png[58 + idat:62 + idat], "big")
    assert png[-12:] == bytes([0, 0, 0, 0, 73, 69, 78, 68, 0xAE, 0x42, 0x60, 0x82])
    if have_ref():
        st, out, ww, hh, cc = ref().decode(png, 4)
        assert st == 0 and (out == img.reshape(-1)).all()


def test_host_buffer_entry_point(enc):
    """fpng_amd_encode_host: what the fpng:: drop-in calls."""
    import fpng_amd
    for (k, w, h, c) in [("grad", 300, 200, 3), ("noise", 64, 64, 4), ("blocks", 777, 33, 4)]:
        img = fpng_amd.synth_image(k, w, h, c)
        for fl in (0, 1, 2):
            ok, png = fpng_amd.fpng_encode_image_to_memory(img, w, h, c, fl)
            assert ok
            _assert_same(png, oracle().encode(img, w, h, c, fl), f"{k} {w}x{h}x{c} f{fl}")
    ok, png = fpng_amd.fpng_encode_image_to_memory(np.zeros(16, dtype=np.uint8), 0, 4, 3)
    assert not ok                                           # reference fpng.cpp:1670-1674
    ok, png = fpng_amd.fpng_encode_image_to_memory(np.zeros(16, dtype=np.uint8), 2, 2, 2)
    assert not ok                                           # reference fpng.cpp:1676-1680


def test_host_batch_overlapped_copies_and_file_writers(enc, tmp_path):
    """fpng_amd_encode_host_batch: more frames than ring slots, mixed sizes, to memory and to files (writer pool)."""
    import fpng_amd
    specs = [("grad", 640, 480, 3), ("blocks", 333, 77, 4), ("noise", 50, 50, 4), ("grad", 1921, 131, 4), ("solid", 1, 1, 3),
             ("grad", 800, 600, 4), ("blocks", 1024, 257, 3), ("grad", 64, 2000, 4), ("noise", 300, 200, 3)]
    imgs = [fpng_amd.synth_image(k, w, h, c, seed=50 + i) for i, (k, w, h, c) in enumerate(specs)]
    for fl in (0, 1):
        exp = [oracle().encode(im, im.shape[1], im.shape[0], im.shape[2], fl) for im in imgs]
        outs = [np.empty(fpng_amd.max_encoded_size(im.shape[1], im.shape[0], im.shape[2]), dtype=np.uint8) for im in imgs]
        sizes = enc.encode_host_batch(imgs, fl, outs=outs)
        for o, n, e in zip(outs, sizes, exp):
            _assert_same(o[:n].tobytes(), e, f"host batch flags={fl}")
        paths = [str(tmp_path / f"f{fl}_{i}.png") for i in range(len(imgs))]
        sizes = enc.encode_host_batch(imgs, fl, paths=paths, writer_threads=3)
        for p, n, e in zip(paths, sizes, exp):
            with open(p, "rb") as f:
                assert f.read() == e and n == len(e)


def test_tickets_return_each_submissions_records(enc):
    """fpng_amd_encode_submit / _wait: several submissions in flight, every one's result records are retrievable."""
    import torch
    import fpng_amd
    subs = []
    for b in range(5):
        imgs = [fpng_amd.synth_image(k, 200 + 40 * b, 100 + i, 4, seed=10 * b + i) for i, k in enumerate(("grad", "blocks", "noise"))]
        ts = [torch.from_numpy(i).cuda() for i in imgs]
        outs = [torch.empty(fpng_amd.max_encoded_size(t.shape[1], t.shape[0], 4) + 64, dtype=torch.uint8, device="cuda") for t in ts]
        enc.submit(ts, outs, 0)
        subs.append((imgs, outs, enc.last_ticket))
    for imgs, outs, ticket in reversed(subs):   # any order
        res = enc.wait(ticket, 3)
        for img, out, (size, mode, status) in zip(imgs, outs, res):
            exp = oracle().encode(img, img.shape[1], img.shape[0], 4, 0)
            assert status == 0 and size == len(exp) and mode == ((exp[60] >> 1) & 3 == 0)
            _assert_same(bytes(out[:size].cpu().numpy()), exp, "ticketed submission")
    enc.finish(3)
    with pytest.raises(Exception):
        enc.wait(10 ** 9, 1)      # unknown ticket


def test_submit_is_ordered_behind_torch_work_without_host_sync(enc):
    """The default Encoder follows torch's current stream: pixels produced by torch kernels immediately before submit()
    (no synchronisation in between) are the pixels that get encoded."""
    import torch
    import fpng_amd
    base = torch.from_numpy(fpng_amd.synth_image("grad", 2048, 1536, 4)).cuda()
    for rep in range(4):
        big = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        big.random_()                                   # keep the stream busy in front of the producer
        img = (base.to(torch.int16) + (rep + 1)).to(torch.uint8)          # the producer of the pixels: torch kernels
        out = torch.empty(fpng_amd.max_encoded_size(2048, 1536, 4) + 64, dtype=torch.uint8, device="cuda")
        enc.submit([img], [out], 0)
        (size, mode, status), = enc.finish(1)
        ref_img = img.cpu().numpy()
        _assert_same(bytes(out[:size].cpu().numpy()), oracle().encode(ref_img, 2048, 1536, 4, 0), f"torch-produced pixels, rep {rep}")


def test_repeated_submissions_reuse_scratch(enc):
    import fpng_amd
    img = fpng_amd.synth_image("grad", 800, 600, 4)
    exp = oracle().encode(img, 800, 600, 4, 0)
    for _ in range(5):
        (png,), _ = _gpu_encode(enc, [img], 0)
        assert png == exp


def test_two_pass_skewed_histogram_quirk(enc):
    """adjust_freq32 with a large skewed histogram: vertical deltas in {0,+1,-1} plus
    ~200 singleton outlier byte values; the 16-bit scaled counts sum past 65535 here."""
    rng = np.random.default_rng(12)
    w, h, c = 1024, 256, 3
    d = rng.choice(np.array([0, 1, 255], dtype=np.uint8), size=(h, w * c), p=[0.9, 0.05, 0.05])
    pos = rng.choice(h * w * c, size=200, replace=False)
    d.reshape(-1)[pos] = rng.choice(np.arange(3, 250), size=200, replace=False).astype(np.uint8)
    img = np.cumsum(d.astype(np.int64), axis=0).astype(np.uint8).reshape(h, w, c)
    (png,), _ = _gpu_encode(enc, [img], 1)
    _assert_same(png, oracle().encode(img, w, h, c, 1), "skewed 2-pass")


def test_row_bands_stitched_at_bit_granularity(enc):
    """The multi-GPU row-band path (fpng_amd_band_hist / _band_encode / _band_place / _band_crc_partials / _wrap_png_crc) driven band after
    band on one GPU: one IDAT, one Deflate block, byte-identical to the whole-image encoding."""
    import torch
    import fpng_amd
    from fpng_amd import sharded
    be = sharded.GpuBandBackend(enc)
    rng = np.random.default_rng(21)
    cases = [fpng_amd.synth_image("grad", 640, 97, 4), fpng_amd.synth_image("blocks", 500, 64, 3),
             fpng_amd.synth_image("grad", 1921, 33, 3), fpng_amd.synth_image("noise", 40, 40, 4)]
    for _ in range(25):
        img, w, h, c = fuzz_image(rng, force_dims=(int(rng.integers(1, 200)), int(rng.integers(2, 30))))
        cases.append(img)
    for img in cases:
        h, w, c = img.shape
        for nb in (2, 3, 8):
            cuts = [0] + sorted(int(v) for v in rng.integers(0, h + 1, nb - 1)) + [h]
            for fl in (0, 1):   # 2-pass: the bands' histograms are summed, every band is coded with the image's table
                png = sharded.encode_image_bands_local(be, torch.from_numpy(np.ascontiguousarray(img)).cuda(), cuts, fl)
                _assert_same(png, oracle().encode(img, w, h, c, fl), f"bands {w}x{h}x{c} cuts={cuts} flags={fl}")


def test_row_bands_4k_eight_bands(enc):
    import torch
    import fpng_amd
    from fpng_amd import sharded
    img = fpng_amd.synth_image("grad", 3840, 2160, 4)
    cuts = [b[0] for b in sharded.split_rows(2160, 8)] + [2160]
    png = sharded.encode_image_bands_local(sharded.GpuBandBackend(enc), torch.from_numpy(img).cuda(), cuts)
    assert hashlib.sha256(png).hexdigest() == "d0f30341ef67c6ea2f67fdb6d6892d493e77999a70b1b0415e16eb81ea653d33"
    png = sharded.encode_image_bands_local(sharded.GpuBandBackend(enc), torch.from_numpy(img).cuda(), cuts, 1)
    assert hashlib.sha256(png).hexdigest() == "69c0ad6dba32822b642e2b4ca5e247c554dd19e08cb848262dce309674a453cd"


def test_row_sharded_through_nccl_one_rank(enc):
    """The real collective path (torch.distributed backend nccl = RCCL) with a one-rank group: all_reduce of the histogram,
    all_gather of the band records, window merge, wrap."""
    import subprocess
    import sys
    code = r'''
import os, sys, hashlib
sys.path.insert(0, os.path.join(os.environ["FPNG_ROOT"], "tests")); sys.path.insert(0, os.environ["FPNG_ROOT"])
import torch, torch.distributed as dist, fpng_amd
from fpng_amd import sharded
from cpu_ref import oracle
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29533"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
enc = fpng_amd.Encoder(device=0)
be = sharded.GpuBandBackend(enc)
for (k, w, h, c) in [("grad", 1921, 257, 4), ("blocks", 640, 480, 3), ("noise", 100, 60, 4)]:
    img = fpng_amd.synth_image(k, w, h, c)
    for fl in (0, 1):
        png = sharded.encode_image_row_sharded(be, torch.from_numpy(img).cuda(), None, w, h, c, 0, h, fl)
        assert bytes(png.cpu().numpy()) == oracle().encode(img, w, h, c, fl), (k, w, h, c, fl)
print("nccl one-rank ok")
dist.destroy_process_group()
'''
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FPNG_ROOT=ROOT), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "nccl one-rank ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_cpp_dropin_namespace_fpng(enc, tmp_path):
    """include/fpng.h + libfpng.so: the reference's own C++ signatures (std::vector out_buf) end to end."""
    import dropin
    import fpng_amd
    L = dropin.shim()
    assert L.shim_supported() == 1
    for (k, w, h, c) in [("grad", 640, 480, 3), ("blocks", 200, 100, 4), ("noise", 31, 17, 4), ("solid", 1, 1, 3)]:
        img = fpng_amd.synth_image(k, w, h, c)
        for fl in (0, 1, 2):
            png = dropin.encode(img, w, h, c, fl)
            _assert_same(png, oracle().encode(img, w, h, c, fl), f"dropin {k} {w}x{h}x{c} f{fl}")
            st, out, *_ = dropin.decode(png, c)
            assert st == 0 and (out == img.reshape(-1)).all()
    assert dropin.encode(np.zeros(12, dtype=np.uint8), 0, 1, 3) is None      # reference fpng.cpp:1670
    assert dropin.encode(np.zeros(12, dtype=np.uint8), 2, 2, 5) is None      # reference fpng.cpp:1676
    img = fpng_amd.synth_image("grad", 100, 50, 4)
    path = str(tmp_path / "x.png").encode()
    assert L.shim_encode_file(path, img.ctypes.data, 100, 50, 4, 0) == 1
    with open(path, "rb") as f:
        assert f.read() == oracle().encode(img, 100, 50, 4, 0)


def test_command_line_harness(tmp_path):
    """fpng_amd_test (tools/fpng_amd_test.cpp, the fpng_test workflow over the drop-in): timing run, CSV, and both fuzz modes
    with the CPU encoders as byte-for-byte judges."""
    import subprocess
    exe = os.path.join(ROOT, "fpng_amd", "lib", "fpng_amd_test")
    judges = [os.path.join(ROOT, "oracle", "libfpng_oracle.so")]
    if have_ref():
        judges.append(os.path.join(ROOT, "oracle", "_ref", "libfpng_ref.so"))
    out = str(tmp_path / "fpng.png")

    def run(*args):
        r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})
        assert r.returncode == 0, (args, r.stdout[-800:], r.stderr[-800:])
        return r.stdout

    for j in judges:
        txt = run("--judge", j, "-o", out, "-b", "4", "-p", "2", "synth:grad:1280x720x4")
        assert "bytes identical" in txt and "device-resident" in txt and "threads, one image each" in txt
        with open(out, "rb") as f:
            import fpng_amd
            assert f.read() == oracle().encode(fpng_amd.synth_image("grad", 1280, 720, 4), 1280, 720, 4, 0)
        csv = run("--judge", j, "-c", "-s", "-o", out, "synth:blocks:640x480x3")
        assert csv.count(",") == 13 and csv.startswith("synth:blocks:640x480x3, 640, 480, 3,")
        assert "trials ok (byte-identical" in run("--judge", j, "-e", "-n", "40", "synth:grad:200x150x4")
        assert "trials ok (byte-identical" in run("--judge", j, "-e", "-s", "-n", "20", "synth:blocks:333x77x3")
        assert "trials ok (byte-identical" in run("--judge", j, "-E", "-n", "6", "-m", "700")
    assert "decode verified" in run("-u", "-o", out, out)      # a file written by fpng as input, no judge: round trip only


def test_command_line_harness_on_ordinary_png_files(tmp_path):
    """The reference's corpus workflow (fpng_test.cpp:1116-1190): ANY PNG in, alpha from -a or from a second file, 24 or 32 bpp
    by content, every output identical to the CPU encoder's; and the six kinds of fuzz damage on a photograph, incl. the byte
    runs that are not pixel-aligned."""
    import subprocess
    import sys
    exe = os.path.join(ROOT, "fpng_amd", "lib", "fpng_amd_test")
    judge = os.path.join(ROOT, "oracle", "_ref", "libfpng_ref.so") if have_ref() else os.path.join(ROOT, "oracle", "libfpng_oracle.so")
    corpus = str(tmp_path / "corpus")
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_corpus.py"), corpus], env=dict(env, FPNG_CORPUS_DROPIN="1"))
    csv = str(tmp_path / "corpus.csv")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "run_corpus.sh"), corpus, csv], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in open(csv).read().splitlines() if ln and not ln.startswith("#")]
    assert r.returncode == 0 and "corpus rc=0" in r.stdout, (r.stdout[-600:], r.stderr[-600:], lines[-3:])
    n_png = len([f for f in os.listdir(corpus) if f.endswith(".png")])  # (photograph-derived files + the screenshot-like content)
    assert n_png >= 17 and len(lines) == 2 * n_png + 2 and not any("FAILED" in ln for ln in lines)
    chans = {ln.split(",")[0].split("/")[-1]: int(ln.split(",")[3]) for ln in lines}
    assert chans["photo_rgba.png"] == 4 and chans["photo_grey.png"] == 3 and chans["photo_palette64.png"] == 3
    assert chans["ui_glyphs_1920x1080x3.png"] == 3 and chans["ui_matte_3840x2160x4.png"] == 4
    assert int(lines[-1].split(",")[3]) == 4 and int(lines[-2].split(",")[3]) == 4   # alpha file / -a: 32 bpp

    def run(*args):
        q = subprocess.run([exe, *args], capture_output=True, text=True, timeout=900, env=env)
        assert q.returncode == 0, (args, q.stdout[-800:], q.stderr[-800:])
        return q.stdout
    photo = os.path.join(corpus, "photo_half.png")
    txt = run("--judge", judge, "-e", "-n", "150", photo)
    assert "trials ok (byte-identical" in txt
    for kind in ("color fill runs", "fill runs", "corrupt runs", "bits flipped"):
        assert kind in txt, kind
    assert "trials ok (byte-identical" in run("--judge", judge, "-e", "-s", "-a", "-n", "60", photo)


def test_pipelined_submissions_without_intermediate_finish(enc):
    """fpng_amd_encode_batch_async() may be called repeatedly before fpng_amd_encode_finish(): submissions
    go through a ring of pinned slots and alternate between the encoder's two lanes."""
    import torch
    import fpng_amd
    rng = np.random.default_rng(55)
    batches, outs_all = [], []
    for b in range(7):   # more than the 4 slots of the ring
        imgs = [fpng_amd.synth_image(k, w, h, c, seed=100 + b) for (k, w, h, c) in
                [("grad", 320 + 16 * b, 200, 4), ("blocks", 257, 64 + b, 3), ("noise", 40, 30, 4)]]
        ts = [torch.from_numpy(i).cuda() for i in imgs]
        outs = [torch.empty(fpng_amd.max_encoded_size(t.shape[1], t.shape[0], t.shape[2]) + 64, dtype=torch.uint8, device="cuda") for t in ts]
        enc.submit(ts, outs, b % 2)      # alternate 1-pass / 2-pass
        batches.append((imgs, ts, b % 2))
        outs_all.append(outs)
    enc.finish(3)
    for (imgs, ts, fl), outs in zip(batches, outs_all):
        for img, out in zip(imgs, outs):
            h, w, c = img.shape
            exp = oracle().encode(img, w, h, c, fl)
            got = bytes(out[:len(exp)].cpu().numpy())
            _assert_same(got, exp, f"pipelined {w}x{h}x{c} flags={fl}")


def test_overlapping_submissions_large_enough_to_run_concurrently(enc):
    """Consecutive submissions run on two internal lanes with separate scratch: make them long enough to
    really overlap on the GPU (several ms each) and check every PNG of every submission."""
    import torch
    import fpng_amd
    subs = []
    for b in range(6):
        c = 4 if b % 3 else 3
        imgs = [fpng_amd.synth_image(kind, 2048, 768 + 64 * b, c, seed=900 + 10 * b + i) for i, kind in enumerate(("grad", "blocks", "noise"))]
        ts = [torch.from_numpy(i).cuda() for i in imgs]
        outs = [torch.empty(fpng_amd.max_encoded_size(t.shape[1], t.shape[0], t.shape[2]) + 64, dtype=torch.uint8, device="cuda") for t in ts]
        subs.append((imgs, ts, outs, 1 if b in (2, 3) else 0))
    torch.cuda.synchronize()
    for imgs, ts, outs, fl in subs:
        enc.submit(ts, outs, fl)
    enc.join()        # device-side join on the encoder's stream, then the host-side wait
    enc.finish(3)
    for imgs, ts, outs, fl in subs:
        for img, out in zip(imgs, outs):
            h, w, c = img.shape
            exp = oracle().encode(img, w, h, c, fl)
            _assert_same(bytes(out[:len(exp)].cpu().numpy()), exp, f"overlapped {w}x{h}x{c} flags={fl}")


def test_scratch_limit_fails_loudly():
    """A submission whose local streams would not fit FPNG_AMD_LOCAL_LIMIT_MB is refused with OUT_OF_MEMORY before anything
    is launched (there is no second, slower path to fall back to)."""
    import subprocess
    import sys
    code = r'''
import sys, os
sys.path.insert(0, os.environ["FPNG_ROOT"])
import torch, fpng_amd
enc = fpng_amd.Encoder(device=0)
img = torch.from_numpy(fpng_amd.synth_image("grad", 2048, 1024, 4)).cuda()
try:
    enc.encode_tensors([img], 0)
except fpng_amd.FpngAmdError as e:
    assert e.code == -5, e.code
    print("refused ok")
'''
    env = dict(os.environ, FPNG_ROOT=ROOT, FPNG_AMD_LOCAL_LIMIT_MB="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "refused ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_wide_rows_fuzz(enc, flags):
    """Rows of 257..4100 pixels built in filtered space from runs of every interesting length, isolated pairs and literal
    stretches at random alignment to the 256-pixel super-windows (tools/gpu_wide_fuzz.py; a 27 000-image run of the same
    generator is in profiles/r03_fuzz_campaign.txt): batches of 200 through one submission each."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_wide_fuzz", os.path.join(ROOT, "tools", "gpu_wide_fuzz.py"))
    wf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(wf)
    rng = np.random.default_rng(4000 + flags)
    for _ in range(3):
        cases = [wf.wide_image(rng) for _ in range(200)]
        pngs, _ = _gpu_encode(enc, [c[0] for c in cases], flags)
        for (img, w, h, c), p in zip(cases, pngs):
            _assert_same(p, oracle().encode(img, w, h, c, flags), f"wide fuzz {w}x{h}x{c}")


def test_flag_bits_the_reference_does_not_know_are_ignored(enc):
    """reference src/fpng.cpp:1662-1803 looks at FPNG_ENCODE_SLOWER and FPNG_FORCE_UNCOMPRESSED only (flags = 3: stored wins); every
    other bit of `flags` must change nothing -- the kernels keep bits of their own in a job's flag word, a caller's bits must never
    reach them.  Device-resident submission and the drop-in's host path."""
    import torch
    import dropin
    import fpng_amd
    rng = np.random.default_rng(808)
    cases = [fuzz_image(rng) for _ in range(6)] + [(fpng_amd.synth_image("grad", 300, 40, 4), 300, 40, 4), (fpng_amd.synth_image("blocks", 129, 33, 3), 129, 33, 3)]
    for img, w, h, c in cases:
        t = torch.from_numpy(np.ascontiguousarray(img).reshape(h, w, c)).cuda()
        for fl in (0, 1, 2, 3):
            want = oracle().encode(img, w, h, c, fl)
            assert want == oracle().encode(img, w, h, c, fl | 0x700) and (not have_ref() or want == ref().encode(img, w, h, c, fl | 0x80000700))
            for high in (0, 0x100, 0x200, 0x400, 0x80000704):
                (png,), _ = enc.encode_tensors([t], fl | high)
                assert png == want, (w, h, c, fl, hex(high))
                assert dropin.encode(img, w, h, c, fl | high) == want, (w, h, c, fl, hex(high), "drop-in")


@pytest.mark.skipif(not have_ref(), reason="the reference decides where the outcome flips")
@pytest.mark.parametrize("shape", [(64, 32, 4), (61, 17, 3), (256, 9, 4), (85, 30, 3), (33, 33, 4), (1024, 3, 3), (1920, 8, 4), (4096, 5, 3)],
                         ids=lambda s: "%dx%dx%d" % s)
def test_stored_or_compressed_at_the_exact_flip_point(enc, shape):
    """The reference falls back to stored blocks when its coder "runs out of buffer": PUT_BITS_FLUSH fails once fewer than 8 bytes
    are left in a buffer sized for the stored form (src/fpng.cpp:567-588, the fallback :1728-1758).  The kernels decide the same
    thing in closed form from the final bit position (scan_kernel).  Images whose first K pixels are noise and the rest flat: K is
    swept over the point where the reference's outcome flips, +-40 pixels, 1-pass and 2-pass -- every file byte-identical, and both
    outcomes present in every sweep.  (Round 4 held this on the CPU for the checker and the band planner only.)"""
    w, h, c = shape
    rng = np.random.default_rng(2718 + w * 31 + h)
    noise = rng.integers(0, 256, (w * h, c), dtype=np.uint8)
    stored = lambda png: (png[60] >> 1) & 3 == 0

    def make(k):
        img = np.full((w * h, c), 77, dtype=np.uint8)
        img[:k] = noise[:k]
        return img.reshape(h, w, c)
    for flags in (0, 1):
        lo, hi = 0, w * h
        assert stored(ref().encode(make(hi), w, h, c, flags)) and not stored(ref().encode(make(0), w, h, c, flags))
        while hi - lo > 1:
            mid = (lo + hi) // 2
            lo, hi = (lo, mid) if stored(ref().encode(make(mid), w, h, c, flags)) else (mid, hi)
        ks = list(range(max(0, hi - 40), min(w * h, hi + 40) + 1))
        imgs = [make(k) for k in ks]
        pngs, _ = _gpu_encode(enc, imgs, flags)
        outcomes = set()
        for k, p, i in zip(ks, pngs, imgs):
            exp = ref().encode(i, w, h, c, flags)
            _assert_same(bytes(p), exp, f"{w}x{h}x{c} flags {flags}, K = {k} (flip at {hi})")
            outcomes.add(stored(exp))
        assert outcomes == {False, True}


@pytest.mark.skipif(not have_ref(), reason="the reference's length limiter is the judge")
def test_length_limited_tables_from_the_device_builder(enc):
    """build_dynamic_kernel on histograms whose optimal prefix code is deeper than fpng's 12 bits (from 13 levels with geometric
    counts on): defl_huffman_enforce_max_code_size (src/fpng.cpp:663-674) decides the table, adjust_freq32 (:909-988) the
    counts it sees.  2-pass files byte-identical to the reference's, 1-pass along the way, and every file decoded back by the GPU
    decoder (dec_build_lut_kernel with 12-bit codes)."""
    from test_oracle import skewed_images
    imgs = skewed_images(np.random.default_rng(1618), max_bytes=3_000_000)
    assert len(imgs) >= 60
    deep = 0
    for flags in (1, 0):
        pngs, _ = _gpu_encode(enc, [i.reshape(h, w, c) for i, w, h, c in imgs], flags)
        for p, (i, w, h, c) in zip(pngs, imgs):
            _assert_same(bytes(p), ref().encode(i, w, h, c, flags), f"skewed histogram {w}x{h}x{c} flags {flags}")
        back = enc.decode_batch(pngs, 4)
        for (st, px, _), (i, w, h, c) in zip(back, imgs):
            assert st == 0 and np.array_equal(px.cpu().numpy()[:, :, :c].reshape(-1), i), (w, h, c, flags, st)
        if flags == 1:  # (the test means what it says only if some tables are at the limit)
            import test_decode_model as M
            for p in pngs:
                res, mode, *_, lut = M.plan(bytes(p))
                if res.status == 0 and mode == 0:  # (the literals' code lengths lie behind the 4096 lookup entries)
                    deep += int(lut[4096:4096 + 64].view(np.uint8).max() == 12)
    assert deep >= 10, deep  # (files whose literal codes reach the 12-bit limit)


def _direct_cases(rng):
    """Images that put seams where they hurt a walk in 256-pixel steps: widths just over / under / on multiples of 256, runs that cross seams or end
    on them, flat rows (one run over the whole row: the look back over the pixels in front of a piece walks to the row's start),
    64-pixel tiles, noise (chunks that overflow the wave's window and spill), gradients."""
    import fpng_amd
    cases = []
    for w in (257, 300, 511, 512, 513, 768, 1025, 1280, 1537, 2049, 2304):
        for c in (3, 4):
            h = int(rng.integers(3, 12))
            for kind in ("grad", "blocks", "solid"):
                cases.append((fpng_amd.synth_image(kind, w, h, c, seed=int(rng.integers(1, 1 << 30))), w, h, c))
            # runs of random lengths (1..700 pixels) of random colours, rows repeated now and then (Up-filtered: all-zero rows)
            img = np.zeros((h, w, c), dtype=np.uint8)
            for y in range(h):
                if y and rng.random() < 0.3:
                    img[y] = img[y - 1]
                    continue
                x = 0
                while x < w:
                    n = int(rng.integers(1, 700)) if rng.random() < 0.5 else int(rng.integers(1, 6))
                    img[y, x:x + n] = rng.integers(0, 256, c, dtype=np.uint8)
                    x += n
            cases.append((img, w, h, c))
            # a run that ends exactly on / one pixel after / one before every multiple of 256
            img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
            for y in range(h):
                for s0 in range(256, w, 256):
                    e = s0 + int(rng.integers(-1, 2))
                    b = max(0, e - int(rng.integers(2, 200)))
                    img[y, b:e] = img[y, b]
            cases.append((img, w, h, c))
    cases.append((fpng_amd.synth_image("noise", 2048, 5, 4), 2048, 5, 4))
    cases.append((fpng_amd.synth_image("noise", 3000, 4, 3), 3000, 4, 3))
    return cases


def test_runs_across_super_window_borders(enc):
    """The row walk takes 256 pixels per step: widths just over / under / on multiples of 256, runs that cross such a border or end
    on it, flat rows, tiles, noise.  (The case list was written for round 5's direct-placement kernel, whose row pieces had their
    seams there; that kernel lost 2.3 x and left the product in round 6 -- profiles/r05_encode_onchip_ab.txt keeps its record --
    the cases stay, through the product's chain.)  Byte-identical to the checker, both passes."""
    rng = np.random.default_rng(5150)
    cases = _direct_cases(rng)
    judge = ref() if have_ref() else oracle()
    for flags in (0, 1):
        for k in range(0, len(cases), 24):
            part = cases[k:k + 24]
            pngs, _ = _gpu_encode(enc, [i for i, *_ in part], flags)
            for p, (img, w, h, c) in zip(pngs, part):
                _assert_same(bytes(p), judge.encode(img, w, h, c, flags), f"{w}x{h}x{c} flags {flags}")


@pytest.mark.parametrize("case,env,expect", [("early", {}, "OK 8 4 library_set"), ("early", {"FPNG_AMD_KEEP_HW_QUEUES": "1"}, "OK 4 2 hands_off"),
                                             ("late", {}, "OK 4 2 driver_open"), ("early", {"GPU_MAX_HW_QUEUES": "16", "FPNG_AMD_LANES": "8"}, "OK 16 8 caller_set")])
def test_lanes_follow_the_hardware_queues(built_lib, case, env, expect):
    """csrc/api.cpp runtime_defaults() / default_lanes(), reported by fpng_amd_runtime_info(): a process that loads the library before
    its first HIP call gets eight hardware queues and four lanes, one that comes too late (a HIP call first: "driver_open") or says
    hands off keeps the runtime's four queues and two lanes;
    in every case seven submissions in flight at once give the checker's files (tests/lanes_check.py)."""
    import subprocess
    import sys
    e = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "FPNG_AMD_KEEP_HW_QUEUES", "FPNG_AMD_LANES")}
    e.update(env, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lanes_check.py"), case], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert [ln for ln in out.stdout.splitlines() if ln.startswith("OK")][-1] == expect


@pytest.mark.parametrize("submissions", [2, 4, 8, 100])
def test_a_batch_in_several_submissions(enc, submissions):
    """Encoder.encode_tensors(..., submissions=k): the batch cut into k submissions whose chains overlap on the lanes (what a caller
    with frames in hand and an idle GPU should do: INTEGRATION.md) -- same files as one submission, both modes, mixed shapes."""
    import torch
    import fpng_amd
    specs = [("grad", 1920, 1080, 4), ("blocks", 640, 480, 3), ("noise", 333, 77, 4), ("grad", 2048, 1536, 3), ("solid", 800, 600, 4),
             ("grad", 4096, 16, 3), ("blocks", 1280, 720, 4), ("grad", 1, 1, 3), ("noise", 64, 64, 3), ("grad", 3840, 2160, 4), ("blocks", 97, 1031, 3)]
    imgs = [fpng_amd.synth_image(k, w, h, c, seed=700 + i) for i, (k, w, h, c) in enumerate(specs)]
    ts = [torch.from_numpy(im).cuda() for im in imgs]
    for flags in (0, 1):
        pngs, _ = enc.encode_tensors(ts, flags, submissions=submissions)
        for (k, w, h, c), im, png in zip(specs, imgs, pngs):
            _assert_same(png, oracle().encode(im, w, h, c, flags), f"{submissions} submissions, {k} {w}x{h}x{c} flags {flags}")
