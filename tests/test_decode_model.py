"""CPU model of the GPU decoder (fpng_amd/csrc/decode.hip) on top of what its host side prepares (fpng_amd_decode_plan: container
checks, block header, the kernels' lookup table): subsequences of 512 token bits decoded speculatively from their nominal first
bits, synchronisation rounds in place until every subsequence starts where its predecessor ended, the stream's end = the FIRST
end-of-block symbol of the chain, output offsets by prefix sums, literals into the filtered image / matches into a run mask, runs
filled from the left, the Up filter undone.  The model must reproduce the pixels; it pins the table format (symbol | length << 9 |
extra bits << 13 | base << 16), the one-32-bit-window-per-token reader and the hand-over rules.  The kernels themselves are held
against the CPU decoder and the reference decoder by tests/test_gpu_decode.py."""
import ctypes as C

import numpy as np
import pytest

from cpu_ref import fuzz_image, oracle

SUB = 512
EOB, INVALID = 1, 4


def plan(png):
    from fpng_amd import _lib
    lib = _lib.load()
    b = np.frombuffer(png, dtype=np.uint8)
    res = _lib.DecodeResult()
    mode, ofs, ln = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    first, limit = C.c_uint64(0), C.c_uint64(0)
    lut = (C.c_uint32 * 4096)()
    rc = lib.fpng_amd_decode_plan(b.ctypes.data, b.size, C.byref(res), C.byref(mode), C.byref(ofs), C.byref(ln), C.byref(first), C.byref(limit), C.byref(lut))
    assert rc == 0
    return res, mode.value, ofs.value, ln.value, first.value, limit.value, np.frombuffer(lut, dtype=np.uint32).copy()


class Model:
    def __init__(self, png):
        self.res, self.mode, ofs, ln, self.first, self.limit, lut = plan(png)
        self.lut = [int(v) for v in lut]
        self.zlen = ln
        z = png[ofs + 8: ofs + 8 + ln] + bytes(16)
        self.zint = int.from_bytes(z, "little")  # LSB-first bit string

    def token(self, pos):
        w = (self.zint >> pos) & 0xFFFFFFFF  # one 32-bit window per token
        e = self.lut[w & 4095]
        ln = (e >> 9) & 15
        if not ln:
            return -1, pos, 0
        sym = e & 511
        if sym <= 256:
            return sym, pos + ln, 0
        xb = (e >> 13) & 7
        return sym, pos + ln + xb + 1, (e >> 16) + ((w >> ln) & ((1 << xb) - 1))

    def decode_sub(self, i, s):
        boundary = self.first + (i + 1) * SUB
        pos, nbytes, fl = s, 0, 0
        while pos < boundary:
            if pos >= self.limit:
                fl = INVALID
                break
            sym, pos, run = self.token(pos)
            if sym < 0:
                fl = INVALID
                break
            if sym == 256:
                fl = EOB
                break
            nbytes += 1 if sym < 256 else run
        return (boundary if fl else pos), nbytes, fl  # a derailed / ended decode hands over on the nominal boundary

    def synchronise(self):
        n = (self.limit - self.first + SUB - 1) // SUB
        start = [self.first + i * SUB for i in range(n)]
        out = [self.decode_sub(i, start[i]) for i in range(n)]
        rounds = 1
        while True:
            # every subsequence looks at its predecessor's end OF THE ROUND BEFORE (the kernel's threads run at the same time; it
            # updates in place, so a thread may also see a newer end: it then settles sooner, never differently)
            ends = [o[0] for o in out]
            changed = False
            for i in range(n):
                s = ends[i - 1] if i else self.first
                if s != start[i]:
                    start[i], out[i], changed = s, self.decode_sub(i, s), True
            rounds += 1
            if not changed:
                break
            assert rounds < 2000
        return start, out, rounds

    def pixels(self, desired):
        w, h, c = self.res.w, self.res.h, self.res.channels_in_file
        bpl, stride = w * c, w * c + 1
        start, out, rounds = self.synchronise()
        last = next(i for i, o in enumerate(out) if o[2] & EOB)  # the stream ends with the FIRST end-of-block symbol of the chain
        assert all(not (out[i][2] & INVALID) for i in range(last + 1))
        offs = np.concatenate([[0], np.cumsum([out[i][1] for i in range(last + 1)])])
        assert offs[-1] == stride * h
        F = np.zeros((h, bpl), dtype=np.uint8)
        run = np.zeros((h, w), dtype=bool)
        saw_eob = False
        for i in range(last + 1):
            pos, o = start[i], int(offs[i])
            boundary = self.first + (i + 1) * SUB
            while pos < boundary:
                assert pos < self.limit
                sym, pos, rl = self.token(pos)
                assert sym >= 0
                if sym == 256:
                    assert o == stride * h and ((pos + 7) >> 3) + 4 == self.zlen
                    saw_eob = True
                    break
                row, col = divmod(o, stride)
                if sym < 256:
                    if col == 0:
                        assert sym == (2 if row else 0)
                    else:
                        F[row, col - 1] = sym
                    o += 1
                else:
                    assert col and (col - 1) % c == 0 and rl % c == 0 and rl and (col - 1) // c + rl // c <= w
                    run[row, (col - 1) // c:(col - 1) // c + rl // c] = True
                    o += rl
        assert saw_eob
        Fp = F.reshape(h, w, c)
        for y in range(h):  # runs: the nearest literal pixel to the left (zeros at the row's start)
            prev = np.zeros(c, dtype=np.uint8)
            for x in range(w):
                if run[y, x]:
                    Fp[y, x] = prev
                else:
                    prev = Fp[y, x]
        px = np.cumsum(Fp.astype(np.uint32), axis=0).astype(np.uint8)  # Up filter undone (bytes, mod 256)
        if desired == 3:
            px = px[:, :, :3]
        elif c == 3:
            px = np.concatenate([px, np.full((h, w, 1), 255, dtype=np.uint8)], axis=2)
        return px, rounds


def test_model_of_the_gpu_decoder_reproduces_the_pixels():
    import fpng_amd
    rng = np.random.default_rng(41)
    cases = [fuzz_image(rng) for _ in range(12)]
    cases += [(fpng_amd.synth_image("grad", 96, 40, 4), 96, 40, 4), (fpng_amd.synth_image("grad", 131, 33, 3), 131, 33, 3),
              (fpng_amd.synth_image("blocks", 200, 70, 4), 200, 70, 4)]
    n_dynamic = 0
    for img, w, h, c in cases:
        for flags in (0, 1):
            png = oracle().encode(img, w, h, c, flags)
            m = Model(png)
            assert m.res.status == 0 and (m.res.w, m.res.h, m.res.channels_in_file) == (w, h, c)
            if m.mode:  # (fell back to stored blocks)
                continue
            n_dynamic += 1
            for desired in (3, 4):
                px, rounds = m.pixels(desired)
                exp = np.asarray(img).reshape(h, w, c)
                exp = exp[:, :, :3] if desired == 3 else (exp if c == 4 else np.concatenate([exp, np.full((h, w, 1), 255, dtype=np.uint8)], axis=2))
                assert np.array_equal(px, exp), (w, h, c, flags, desired)
    assert n_dynamic >= 12


def test_rounds_settle_quickly_with_and_without_noise():
    """Synchronisation rounds of the model (every subsequence looks at its predecessor's end of the round before) on a gradient with
    and without its noise bits (seed 0: the generator's xorshift stays 0 -- long exact runs): a handful of rounds each at this size.
    (On the GPU an 8K frame of the noise-free kind needed 38 rounds -- profiles/r03_g_decode.txt -- which is why
    fpng_amd_decode_batch gives a group of files more rounds when the six it launches blind were not enough.)"""
    import fpng_amd
    for seed in (0, 12345):
        for (w, h) in ((640, 24), (2048, 8)):
            img = fpng_amd.synth_image("grad", w, h, 4, seed=seed)
            m = Model(oracle().encode(img, w, h, 4, 0))
            px, rounds = m.pixels(4)
            assert np.array_equal(px, img)
            assert rounds <= 8


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
