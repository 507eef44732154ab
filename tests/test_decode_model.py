"""CPU tests of the GPU decoder's logic (fpng_amd/csrc/decode.hip) without a GPU:

* tests/cpp/decode_emul.cpp runs the kernels' own per-thread code (fpng_amd/csrc/decode_core.h) thread by thread on top of what the
  decoder's host side prepares (fpng_amd_decode_plan: container checks, block header, the lookup table): subsequences of 512 token
  bits decoded from a lead-in in front of their nominal first bits, corrections inside a workgroup until every subsequence starts
  where its predecessor ended, rounds across workgroup borders, the stream's end = the FIRST end-of-block symbol of the chain,
  output offsets, the literal bytes in front of every subsequence, the token RECORDS every settling decode leaves, the windows of
  the filtered stream filled from them (every store of every walk checked against the walk's territory, long matches marked and
  filled, resume points, every form of the walk's step), the Up filter undone.  Workgroup size and lead-in are parameters there:
  small ones put many borders and seams into small images.
  It must reproduce the pixels, and the status codes of the CPU decoder / the reference on damaged files.
* a few lines of Python decode a stream token by token with the same table: that pins the table's FORMAT (a simple token: all its
  bits << 28 | literal count << 26 | up to three literal bytes, or bit 25 | match length; other tokens: code bits << 12, a match's
  base length and extra bit count, bit 24 for the end of the block; then the literals' code lengths; token_mutator.entry_code_bits).

The kernels themselves are held against the CPU decoder and the reference decoder by tests/test_gpu_decode.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import dropin
import token_mutator
from cpu_ref import fuzz_image, have_ref, oracle, ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUT_WORDS = 4160
UNDECIDED = 64
# (workgroup size, lead-in bits, tile bytes): the kernels' own constants first
CONFIGS = [(512, 128, 18432), (4, 128, 64), (3, 0, 100), (8, 32, 52), (2, 64, 4), (64, 128, 1024)]

_emul = {}


def emul(variant=""):
    """the emulator library (variant: extra -D options by name, none at present)"""
    if variant not in _emul:
        from fpng_amd import build
        build.build()
        lib_dir = os.path.join(ROOT, "fpng_amd", "lib")
        src = os.path.join(ROOT, "tests", "cpp", "decode_emul.cpp")
        so = os.path.join(lib_dir, f"libfpng_decode_emul{'_' + variant if variant else ''}.so")
        deps = [src, os.path.join(ROOT, "fpng_amd", "csrc", "decode_core.h"), os.path.join(lib_dir, "libfpng_amd.so")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            defs = {"": []}[variant]
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas"] + defs + ["-I", os.path.join(ROOT, "include"), "-I",
                                   os.path.join(ROOT, "fpng_amd", "csrc"), src, "-o", so, "-L", lib_dir, "-lfpng_amd", "-Wl,-rpath,$ORIGIN"])
        L = C.CDLL(so)
        L.fpng_emul_decode.restype = C.c_int
        L.fpng_emul_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t] + [C.POINTER(C.c_uint32)] * 3 + [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)]
        _emul[variant] = L
    return _emul[variant]


def emul_decode(png, desired, cfg=CONFIGS[0], border_rounds=1000, variant=""):
    """-> (status, pixels or None, w, h, c, stats)"""
    b = np.frombuffer(bytes(png), dtype=np.uint8)
    w, h, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    st, ww, hh, cc = dropin.get_info(png)
    cap = ww * hh * desired + 16 if st == 0 else 16
    out = np.zeros(cap, dtype=np.uint8)
    stats = (C.c_uint32 * 4)()
    st = emul(variant or os.environ.get("FPNG_EMUL_VARIANT", "")).fpng_emul_decode(b.ctypes.data, b.size, desired, out.ctypes.data, cap, C.byref(w), C.byref(h), C.byref(c), cfg[0], cfg[1], cfg[2], border_rounds, stats)
    assert st > -1000, f"emulator internal error {st}"
    return st, (out[: w.value * h.value * desired] if st == 0 else None), w.value, h.value, c.value, list(stats)


def expected_pixels(img, w, h, c, desired):
    exp = np.asarray(img).reshape(h, w, c)
    if desired == 3:
        return exp[:, :, :3].reshape(-1)
    return (exp if c == 4 else np.concatenate([exp, np.full((h, w, 1), 255, dtype=np.uint8)], axis=2)).reshape(-1)


def plan(png):
    from fpng_amd import _lib
    lib = _lib.load()
    b = np.frombuffer(png, dtype=np.uint8)
    res = _lib.DecodeResult()
    mode, ofs, ln = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    first, limit = C.c_uint64(0), C.c_uint64(0)
    lut = (C.c_uint32 * LUT_WORDS)()
    rc = lib.fpng_amd_decode_plan(b.ctypes.data, b.size, C.byref(res), C.byref(mode), C.byref(ofs), C.byref(ln), C.byref(first), C.byref(limit), C.byref(lut))
    assert rc == 0
    return res, mode.value, ofs.value, ln.value, first.value, limit.value, np.frombuffer(lut, dtype=np.uint32).copy()


def test_emulated_kernels_reproduce_the_pixels():
    import fpng_amd
    rng = np.random.default_rng(41)
    cases = [fuzz_image(rng) for _ in range(60)]
    cases += [fuzz_image(rng, force_dims=(int(rng.integers(200, 700)), int(rng.integers(8, 30)))) for _ in range(6)]
    cases += [(fpng_amd.synth_image("grad", 96, 40, 4), 96, 40, 4), (fpng_amd.synth_image("grad", 131, 33, 3), 131, 33, 3),
              (fpng_amd.synth_image("blocks", 200, 70, 4), 200, 70, 4), (fpng_amd.synth_image("solid", 300, 9, 3), 300, 9, 3),
              (fpng_amd.synth_image("solid", 1000, 5, 4), 1000, 5, 4)]
    n_dynamic = corrected = border = 0
    for k, (img, w, h, c) in enumerate(cases):
        for flags in (0, 1, 2):
            png = oracle().encode(img, w, h, c, flags)
            res, mode, *_ = plan(png)
            assert res.status == 0 and (res.w, res.h, res.channels_in_file) == (w, h, c)
            n_dynamic += not mode
            for cfg in (CONFIGS if not mode else CONFIGS[:1]):
                for desired in (3, 4):
                    st, px, ww, hh, cc, stats = emul_decode(png, desired, cfg)
                    assert st == 0 and (ww, hh, cc) == (w, h, c), (k, w, h, c, flags, cfg, st)
                    assert np.array_equal(px, expected_pixels(img, w, h, c, desired)), (k, w, h, c, flags, desired, cfg)
                corrected += stats[3]
                border += stats[1]
    assert n_dynamic >= 60
    assert corrected > 0 and border > 0  # (both kinds of correction happened somewhere)


def test_emulated_kernels_on_a_natural_image():
    import real_image
    png = real_image.fixture_bytes()
    st, exp, w, h, c = dropin.decode(png, 3)
    assert st == 0
    for cfg in (CONFIGS[0], (16, 128, 4096), (512, 0, 18432)):
        st, px, ww, hh, cc, stats = emul_decode(png, 3, cfg)
        assert st == 0 and np.array_equal(px, exp)
        # a photograph: with the kernels' lead-in about 2 % of the subsequences start out of step; all of them when there is none
        frac = stats[3] / stats[2]
        assert (frac < 0.06) if cfg[1] == 128 else (frac > 0.5)


def test_correction_rounds_stay_few():
    """Rounds inside a workgroup and across the borders on a gradient with and without its noise bits (seed 0: the generator's xorshift
    stays 0 -- long exact runs, a PERIODIC token stream that keeps a decoder that started at a wrong bit out of step for many
    subsequences): the in-workgroup rounds walk along such a stretch one subsequence at a time, the border rounds hand it over."""
    import fpng_amd
    for seed in (0, 12345):
        for (w, h) in ((640, 24), (2048, 8)):
            img = fpng_amd.synth_image("grad", w, h, 4, seed=seed)
            png = oracle().encode(img, w, h, 4, 0)
            for cfg in (CONFIGS[0], (8, 128, 256)):
                st, px, *_, stats = emul_decode(png, 4, cfg)
                assert st == 0 and np.array_equal(px, img.reshape(-1))
                assert stats[0] <= 40 and stats[1] <= 12, (seed, w, h, cfg, stats)


def periodic_images():
    """Content whose token stream repeats itself: vertical stripes on a vertical ramp (every filtered row the same few bytes), tiles."""
    import fpng_amd
    out = []
    for period, c in ((2, 4), (3, 4), (5, 3), (8, 4), (64, 3)):
        w, h = 4096, 320
        pal = np.random.default_rng(period).integers(0, 256, (period, c), dtype=np.uint8)
        img = (pal[np.arange(w) % period][None] + (np.arange(h)[:, None, None] * 7).astype(np.uint8)).astype(np.uint8)
        out.append((f"stripes{period}", np.ascontiguousarray(img).reshape(-1), w, h, c))
    out.append(("blocks", np.asarray(fpng_amd.synth_image("blocks", 4096, 2048, 4)).reshape(-1), 4096, 2048, 4))
    return out


def test_periodic_streams_do_not_crawl():
    """A stream that repeats itself keeps wrongly started decoders in a stable false phase; without the candidate lists
    (decode_core.h) the corrections walk through a workgroup one subsequence per step (512 steps) and over the borders one
    workgroup per round.  With them: a handful of steps, one border round, whatever the workgroup size."""
    worst = [0, 0]
    crawled = 0
    for name, img, w, h, c in periodic_images():
        for flags in (0, 1):
            png = oracle().encode(img, w, h, c, flags)
            for cfg in (CONFIGS[0], (64, 128, 1024), (16, 128, 256)):
                st, px, ww, hh, cc, stats = emul_decode(png, c, cfg)
                assert st == 0 and np.array_equal(px, img), (name, flags, cfg)
                assert stats[0] <= 12 and stats[1] <= 2, (name, flags, cfg, stats)
                worst = [max(worst[0], stats[0]), max(worst[1], stats[1])]
                crawled += stats[0] > 3  # (more than kRefixRounds steps: the lists were used)
    assert crawled >= 5, worst


def test_phase_map_helpers():
    """decode_core.h's phase maps (18 phases -> phase, three dwords): set / at round trips, composition against its definition,
    associativity (what the prefix "sums" of dec_sync_kernel and dec_chain_kernel rest on) -- 20 000 random maps."""
    L = emul()
    L.fpng_emul_phase_map_selftest.restype = C.c_int
    L.fpng_emul_phase_map_selftest.argtypes = [C.c_uint32, C.c_uint32]
    assert L.fpng_emul_phase_map_selftest(7, 20000) == 0


def test_too_few_border_rounds_leave_the_file_undecided():
    import fpng_amd
    img = fpng_amd.synth_image("grad", 640, 40, 4)
    png = oracle().encode(img, 640, 40, 4, 0)
    st, px, *_, stats = emul_decode(png, 4, (4, 0, 64), border_rounds=1000)
    assert st == 0 and stats[1] >= 1
    st, px, *_ = emul_decode(png, 4, (4, 0, 64), border_rounds=0)
    assert st == UNDECIDED and px is None


def _damage(rng, png):
    bad = bytearray(png)
    kind = int(rng.integers(0, 5))
    if kind == 0:      # flip a bit anywhere
        i = int(rng.integers(0, len(bad)))
        bad[i] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:    # truncate
        bad = bad[: int(rng.integers(1, len(bad)))]
    elif kind == 2:    # damage the block header region
        i = int(rng.integers(58, min(140, len(bad))))
        bad[i] = int(rng.integers(0, 256))
    elif kind == 3:    # damage the zlib stream
        i = int(rng.integers(58, len(bad)))
        bad[i] = int(rng.integers(0, 256))
    else:              # flip a bit in the token bits
        i = int(rng.integers(min(125, len(bad) - 1), len(bad)))
        bad[i] ^= 1 << int(rng.integers(0, 8))
    return kind, bytes(bad)


def test_damaged_files_get_the_cpu_decoders_status():
    """The judge is the reference's decoder when its build is present (dev container, GPU box), the drop-in's CPU decoder otherwise
    (itself held against the reference by tests/test_dropin_decode.py)."""
    judge = ref().decode if have_ref() else dropin.decode
    rng = np.random.default_rng(43)
    checked = rejected = 0
    for _ in range(50):
        img, w, h, c = fuzz_image(rng) if rng.random() < 0.7 else fuzz_image(rng, force_dims=(int(rng.integers(100, 400)), int(rng.integers(4, 12))))
        png = oracle().encode(img, w, h, c, int(rng.integers(0, 2)))
        for _ in range(10):
            kind, bad = _damage(rng, png)
            cfg = CONFIGS[int(rng.integers(0, len(CONFIGS)))]
            for desired in (3, 4):
                st_r, out_r, *_ = judge(bad, desired)
                st_m, out_m, *_ = emul_decode(bad, desired, cfg)
                if st_m == UNDECIDED:  # (left to the CPU decoder, as fpng::fpng_decode_memory does: a match at a row's first pixel)
                    st_m, out_m, *_ = dropin.decode(bad, desired)
                checked += 1
                rejected += st_r != 0
                assert st_m == st_r, (w, h, c, kind, cfg, st_r, st_m)
                if st_r == 0:
                    assert np.array_equal(np.asarray(out_r)[: out_m.size], out_m), (w, h, c, kind, cfg)
    assert checked >= 900 and rejected >= 300


def test_a_match_at_a_rows_first_pixel_is_left_to_the_cpu_decoder():
    """tests/golden/first_pixel_match.png: a 2-pass file of a 65 x 16 RGBA image with ONE flipped bit (found by tools/emul_campaign.py)
    that turns the first token of row 15 into a match -- it repeats a pixel of zeros there (reference src/fpng.cpp:2268: the
    previous-pixel deltas start at 0 in every row), which no fpng encoder writes and the kernels' "last literal bytes" know nothing
    about.  The reference decodes the file; the kernels' logic must say UNDECIDED (kEmitLeaveToCpu), never NOT_FPNG, and the
    drop-in's CPU decoder, which takes over then, must give the reference's pixels."""
    bad = open(os.path.join(ROOT, "tests", "golden", "first_pixel_match.png"), "rb").read()
    for cfg in CONFIGS:
        for desired in (3, 4):
            st, px, *_ = emul_decode(bad, desired, cfg)
            assert st == UNDECIDED and px is None, (cfg, desired, st)
    for desired in (3, 4):
        st_c, out_c, w, h, c = dropin.decode(bad, desired)
        assert st_c == 0 and (w, h, c) == (65, 16, 4)
        if have_ref():
            st_r, out_r, *_ = ref().decode(bad, desired)
            assert st_r == 0 and np.array_equal(np.asarray(out_r)[: out_c.size], out_c)


def edited_files(rng, n_images, per_image=8):
    """[(name of the edit, file)]: valid code streams whose TOKENS were edited (tests/token_mutator.py), raw and with the byte count
    made good again"""
    import token_mutator as TM
    out = []
    for _ in range(n_images):
        img, w, h, c = fuzz_image(rng) if rng.random() < 0.6 else fuzz_image(rng, force_dims=(int(rng.integers(20, 300)), int(rng.integers(2, 12))))
        png = oracle().encode(img, w, h, c, int(rng.integers(0, 2)))
        if plan(png)[1]:
            continue  # (stored blocks: no tokens)
        s = TM.Stream(png, plan)
        same = s.write(s.tokens)
        assert same is not None and dropin.decode(same, c)[0] == 0
        for _ in range(per_image):
            T, name = TM.mutate(s, rng)
            if name == "none":
                continue
            for f in (s.write(T), s.write(TM.balanced(s, T, rng))):
                if f is not None:
                    out.append((name, f))
    return out


def test_edited_token_streams_get_the_references_answer():
    """Flipped bits derail a Huffman stream; these files are VALID code streams that bend or break the decoder's semantic rules
    (reference src/fpng.cpp:2255-2330): matches lengthened / shortened by pixels or by bytes, split, merged, put where no pixel
    starts, at a row's first pixel, up to and over the row's end, the other distance bit, filter literals changed, tokens doubled
    or dropped, the end-of-block symbol early or late.  The reference's decoder judges; the kernels' logic (the CPU decoder where it
    says UNDECIDED) and the drop-in's CPU decoder must give its status and its pixels."""
    if not have_ref():
        pytest.skip("the reference's decoder is the judge")
    rng = np.random.default_rng(2024)
    seen, accepted, left = {}, 0, 0
    for name, f in edited_files(rng, 70):
        cfg = CONFIGS[int(rng.integers(0, len(CONFIGS)))]
        desired = int(rng.choice([3, 4]))
        st_r, out_r, *_ = ref().decode(f, desired)
        st_c, out_c, *_ = dropin.decode(f, desired)
        st_m, out_m, *_ = emul_decode(f, desired, cfg)
        if st_m == UNDECIDED:
            left += 1
            st_m, out_m = st_c, out_c
        for st, o in ((st_m, out_m), (st_c, out_c)):
            assert st == st_r, (name, cfg, desired, st_r, st_m, st_c)
            assert st_r != 0 or np.array_equal(np.asarray(out_r)[: o.size], o), (name, cfg, desired)
        seen[name.split("+")[0].split("-")[0]] = 1
        accepted += st_r == 0
    assert len(seen) >= 12 and accepted >= 200 and left >= 10, (sorted(seen), accepted, left)


def test_edited_token_streams_of_megapixel_images():
    """The same kind of edits, ONE at a random place of a stream that spans dozens of workgroups of the kernels' own size (512
    subsequences of 512 bits): a pixel of literals turned into a match (also at a row's first pixel: left to the CPU decoder),
    matches split, merged, a pixel longer or shorter with the byte count made good, filter literals and pixels changed, matches
    over the row's end or off the pixel grid.  The reference's decoder judges the kernels' logic and the drop-in's CPU decoder."""
    if not have_ref():
        pytest.skip("the reference's decoder is the judge")
    import fpng_amd
    import token_mutator as TM
    import ui_images
    rng = np.random.default_rng(4242)
    imgs = [(np.asarray(fpng_amd.synth_image("grad", 1500, 700, 4)).reshape(-1), 1500, 700, 4),
            (np.ascontiguousarray(ui_images.glyphs(1280, 600, 3, seed=5)).reshape(-1), 1280, 600, 3),
            (np.asarray(fpng_amd.synth_image("blocks", 1100, 900, 3)).reshape(-1), 1100, 900, 3)]
    seen, accepted, left = set(), 0, 0
    for k, (img, w, h, c) in enumerate(imgs):
        png = oracle().encode(img, w, h, c, k % 2)
        s = TM.LargeStream(png, plan, emul())
        assert ref().decode(s.splice(0, 0, []), c)[0] == 0 and s.n > 10000
        for _ in range(14):
            name, f = TM.mutate_large(s, rng)
            if f is None:
                continue
            desired = int(rng.choice([3, 4]))
            st_r, out_r, *_ = ref().decode(f, desired)
            st_c, out_c, *_ = dropin.decode(f, desired)
            st_m, out_m, *_ = emul_decode(f, desired)
            if st_m == UNDECIDED:
                left += 1
                st_m, out_m = st_c, out_c
            for st, o in ((st_m, out_m), (st_c, out_c)):
                assert st == st_r and (st_r != 0 or np.array_equal(np.asarray(out_r)[: o.size], o)), (name, w, h, c, st_r, st_m, st_c)
            seen.add(name)
            accepted += st_r == 0
    assert len(seen) >= 7 and accepted >= 15 and left >= 2, (sorted(seen), accepted, left)


# ---- the table's format, pinned by a decoder of a dozen lines ----
def _serial_decode(png):
    res, mode, ofs, ln, first, limit, lut = plan(png)
    assert res.status == 0 and mode == 0
    lenof = lut[4096:].view(np.uint8)
    zint = int.from_bytes(png[ofs + 8: ofs + 8 + ln] + bytes(16), "little")  # LSB-first bit string
    pos, out, prev_groups = first, bytearray(), 0
    while True:
        assert pos < limit
        wnd = (zint >> pos) & 0xFFFFFFFF
        e = int(lut[wnd & 4095])
        L, n = token_mutator.entry_code_bits(e), (e >> 26) & 3
        assert L, "no such code"
        if n:
            lits = [(e >> (8 * k)) & 255 for k in range(n)]
            assert sum(int(lenof[b]) for b in lits) == L  # the group's bits = its literals' code lengths
            out += bytes(lits)
            pos += L
            prev_groups += n > 1
        elif e & (1 << 25):
            xb, base = (e >> 9) & 7, e & 511
            run = base + ((wnd >> L) & ((1 << xb) - 1))
            out += bytes(out[-1:]) * 0 + bytes(run)  # (placeholder bytes: only the count matters here)
            pos += L + xb + 1
        else:
            pos += L
            break
    assert ((pos + 7) >> 3) + 4 == ln
    return res, len(out), prev_groups


def test_table_format():
    import fpng_amd
    for (kind, w, h, c) in (("grad", 96, 40, 4), ("grad", 131, 33, 3), ("blocks", 200, 70, 4)):
        for flags in (0, 1):
            img = fpng_amd.synth_image(kind, w, h, c)
            res, nbytes, groups = _serial_decode(oracle().encode(img, w, h, c, flags))
            assert nbytes == (w * c + 1) * h
            assert kind != "grad" or groups > 100  # (several literals per lookup are the rule on such content)


def test_plan_reports_container_and_stream_problems():
    import fpng_amd
    img = fpng_amd.synth_image("grad", 64, 20, 3)
    png = oracle().encode(img, 64, 20, 3, 0)
    assert plan(png)[0].status == 0
    bad = bytearray(png)
    bad[20] ^= 0x10  # IHDR payload: header CRC
    assert plan(bytes(bad))[0].status != 0
    bad = bytearray(png)
    bad[60] ^= 0x06  # block type bits
    assert plan(bytes(bad))[0].status != 0
    stored = oracle().encode(img, 64, 20, 3, 2)
    res, mode, *_ = plan(stored)
    assert res.status == 0 and mode == 1
    # a header that promises more pixels than the IDAT could ever hold is turned away before any memory is sized by it
    # (2 bits per token at least, 258 bytes per token at most)
    import struct
    import zlib
    big = bytearray(png)
    big[16:24] = struct.pack(">II", 30000, 30000)
    big[29:33] = struct.pack(">I", zlib.crc32(bytes(big[12:29])))
    assert plan(bytes(big))[0].status == 1  # FPNG_DECODE_NOT_FPNG


def test_periodic_streams_with_many_phases():
    """Random periodic images -- periods 1..39, tiles, a few noisy pixels that break the period: some of their streams keep seven
    and more decoder phases alive (a first version of the phase maps held six pairs and walked through those).  Pixels exact,
    a handful of steps."""
    rng = np.random.default_rng(99)
    worst = 0
    for trial in range(48):
        period, c, w, h = int(rng.integers(1, 40)), int(rng.choice([3, 4])), int(rng.integers(600, 4200)), int(rng.integers(40, 200))
        mode = int(rng.integers(0, 4))
        pal = rng.integers(0, 256, (period, c), dtype=np.uint8)
        if mode == 1:
            pal &= 0xF0
        img = (pal[np.arange(w) % period][None] + (np.arange(h)[:, None, None] * int(rng.integers(0, 9))).astype(np.uint8)).astype(np.uint8)
        if mode == 2:
            img = img[(np.arange(h) // 8 * 8) % h]
        if mode == 3:
            img.reshape(-1, c)[rng.integers(0, w * h, 20)] = rng.integers(0, 256, (20, c), dtype=np.uint8)
        img = np.ascontiguousarray(img).reshape(-1)
        for flags in (0, 1):
            png = oracle().encode(img, w, h, c, flags)
            for cfg in (CONFIGS[0], (64, 128, 1024)):
                st, px, *_, stats = emul_decode(png, c, cfg)
                assert st == 0 and np.array_equal(px, img), (trial, period, c, w, h, mode, flags, cfg)
                assert stats[0] <= 12 and stats[1] <= 3, (trial, period, c, w, h, mode, flags, cfg, stats)
                worst = max(worst, stats[0])
    assert worst > kRefixRounds  # (the maps were needed somewhere)


kRefixRounds = 3
