"""Token-level mutations of fpng files (TEST INFRASTRUCTURE).  A flipped bit nearly always derails a Huffman stream, so random damage
rarely reaches the decoder's SEMANTIC rules (reference src/fpng.cpp:2255-2330: matches are whole pixels, start on a pixel, stay in
their row, repeat the previous pixel -- a pixel of zeros at a row's start --, the filter literals are 0 then 2, the stream ends
exactly with the image).  Here a valid file's token stream is taken apart with the decoder's own lookup table
(fpng_amd_decode_plan), edited token by token, and written back with the same Huffman code: a VALID code stream that bends or
breaks those rules.  The reference's decoder is the judge of what such a file means."""
import struct
import zlib

import numpy as np



def entry_code_bits(e):
    """Code bits of the token a table entry stands for (fpng_amd/csrc/decode_core.h): a simple token -- a group of literals, a match
    without extra bits -- holds ALL the bits it takes in its upper four (a match's 1-bit distance code included), every other token
    its code bits in bits 15..12; 0: no such code."""
    adv, n = e >> 28, (e >> 26) & 3
    if n:
        return adv
    if e & (1 << 25):
        return adv - 1 if adv else (e >> 12) & 15
    return (e >> 12) & 15


class Stream:
    """tokens of a dynamic-block fpng file: ('lit', byte) | ('match', length, dist_bit) | ('eob',)"""

    def __init__(self, png, plan):
        res, mode, ofs, ln, first, limit, lut = plan(png)
        assert res.status == 0 and mode == 0
        self.png, self.ofs, self.ln, self.first = bytes(png), ofs, ln, first
        self.w, self.h, self.c = res.w, res.h, res.channels_in_file
        self.stride = self.w * self.c + 1
        lenof = lut[4096:].view(np.uint8)
        self.lit_code, self.len_syms, self.eob = {}, {}, None
        for i in range(4096):
            e = int(lut[i])
            L, n = entry_code_bits(e), (e >> 26) & 3
            if not L:
                continue
            if n:
                b = e & 255
                l = int(lenof[b])
                self.lit_code[b] = (i & ((1 << l) - 1), l)
            elif e & (1 << 25):
                self.len_syms[(e & 511, (e >> 9) & 7)] = (i & ((1 << L) - 1), L)
            else:
                self.eob = (i & ((1 << L) - 1), L)
        z = self.png[ofs + 8: ofs + 8 + ln]
        self.zint = int.from_bytes(z + bytes(16), "little")
        self.tokens = []
        pos = first
        while True:
            assert pos < limit
            wnd = (self.zint >> pos) & 0xFFFFFFFF
            e = int(lut[wnd & 4095])
            L, n = entry_code_bits(e), (e >> 26) & 3
            assert L
            if n:
                b = e & 255
                self.tokens.append(("lit", b))
                pos += int(lenof[b])
            elif e & (1 << 25):
                xb, base = (e >> 9) & 7, e & 511
                self.tokens.append(("match", base + ((wnd >> L) & ((1 << xb) - 1)), (wnd >> (L + xb)) & 1))
                pos += L + xb + 1
            else:
                self.tokens.append(("eob",))
                break

    def can_match(self, length):
        return any(base <= length < base + (1 << xb) for (base, xb) in self.len_syms)

    def write(self, tokens):
        """-> a PNG with this token list behind the original block header (None if a token has no code in this file's table)"""
        bits, n = self.zint & ((1 << self.first) - 1), self.first
        for t in tokens:
            if t[0] == "lit":
                if t[1] not in self.lit_code:
                    return None
                code, l = self.lit_code[t[1]]
                bits |= code << n
                n += l
            elif t[0] == "match":
                sym = [(base, xb) for (base, xb) in self.len_syms if base <= t[1] < base + (1 << xb)]
                if not sym:
                    return None
                base, xb = sym[0]
                code, l = self.len_syms[(base, xb)]
                bits |= code << n
                n += l
                bits |= (t[1] - base) << n
                n += xb
                bits |= (t[2] & 1) << n
                n += 1
            else:
                if self.eob is None:
                    return None
                bits |= self.eob[0] << n
                n += self.eob[1]
        nbytes = (n + 7) // 8
        z = bits.to_bytes(nbytes, "little") + b"\x12\x34\x56\x78"  # (the Adler-32 is not checked: reference src/fpng.cpp:3016-3026)
        head = self.png[: self.ofs]
        return head + struct.pack(">I", len(z)) + b"IDAT" + z + struct.pack(">I", zlib.crc32(b"IDAT" + z)) + struct.pack(">I", 0) + b"IEND" + struct.pack(">I", zlib.crc32(b"IEND"))

    def positions(self, tokens=None):
        """output byte offset in front of every token"""
        out, p = [], 0
        for t in (tokens if tokens is not None else self.tokens):
            out.append(p)
            p += 1 if t[0] == "lit" else (t[1] if t[0] == "match" else 0)
        return out


def mutate(s, rng):
    """one edited token list (a copy) and a short name of the edit"""
    T = list(s.tokens)
    pos = s.positions(T)
    C, stride = s.c, s.stride
    matches = [i for i, t in enumerate(T) if t[0] == "match"]
    lits = [i for i, t in enumerate(T) if t[0] == "lit"]
    kind = int(rng.integers(0, 12))
    name = "none"
    if kind == 0 and matches:  # a match longer / shorter by whole pixels, the difference made up with the next / previous literal pixels' worth
        i = matches[int(rng.integers(0, len(matches)))]
        d = int(rng.choice([-2, -1, 1, 2])) * C
        if 3 <= T[i][1] + d <= 258:
            T[i] = ("match", T[i][1] + d, T[i][2])
            name = f"match{d:+d}"
    elif kind == 1 and matches:  # ... by a byte count that is no whole pixel
        i = matches[int(rng.integers(0, len(matches)))]
        d = int(rng.choice([-1, 1, 2, -2, 5]))
        if 3 <= T[i][1] + d <= 258:
            T[i] = ("match", T[i][1] + d, T[i][2])
            name = f"matchbytes{d:+d}"
    elif kind == 2 and matches:  # one match split in two (same bytes)
        i = matches[int(rng.integers(0, len(matches)))]
        if T[i][1] >= 2 * max(C, 3):
            a = int(rng.integers(1, T[i][1] // C)) * C
            if a >= 3 and T[i][1] - a >= 3:
                T[i:i + 1] = [("match", a, T[i][2]), ("match", T[i][1] - a, T[i][2])]
                name = "split"
    elif kind == 3 and len(matches) >= 2:  # two neighbouring matches merged
        for i in matches:
            if i + 1 < len(T) and T[i + 1][0] == "match" and T[i][1] + T[i + 1][1] <= 258:
                T[i:i + 2] = [("match", T[i][1] + T[i + 1][1], T[i][2])]
                name = "merge"
                break
    elif kind == 4 and lits:  # C literals that start a pixel replaced by a match of one pixel (also at a row's FIRST pixel)
        cand = [i for i in lits if (pos[i] % stride) >= 1 and (pos[i] % stride - 1) % C == 0 and i + C <= len(T) and all(T[i + k][0] == "lit" for k in range(C))]
        if rng.random() < 0.5:
            first = [i for i in cand if pos[i] % stride == 1]
            cand = first or cand
        if cand and C >= 3:
            i = cand[int(rng.integers(0, len(cand)))]
            T[i:i + C] = [("match", C, 0)]
            name = "lit2match" + ("_firstpx" if pos[i] % stride == 1 else "")
    elif kind == 5 and lits:  # a match where no pixel starts
        i = lits[int(rng.integers(0, len(lits)))]
        if i + 3 <= len(T) and all(T[i + k][0] == "lit" for k in range(3)):
            T[i:i + 3] = [("match", 3, 0)]
            name = "lit2match_anywhere"
    elif kind == 6:  # the end-of-block symbol earlier / more tokens behind the image
        if rng.random() < 0.5 and len(T) > 3:
            cut = int(rng.integers(1, len(T) - 1))
            T = T[:cut] + [("eob",)]
            name = "early_eob"
        else:
            extra = [("lit", int(rng.integers(0, 256))) for _ in range(int(rng.integers(1, 6)))]
            T = T[:-1] + extra + [("eob",)]
            name = "late_eob"
    elif kind == 7:  # a filter literal changed
        rows = [i for i, t in enumerate(T) if t[0] == "lit" and pos[i] % stride == 0]
        if rows:
            i = rows[int(rng.integers(0, len(rows)))]
            T[i] = ("lit", int(rng.choice([0, 1, 2, 3, 4, 255])))
            name = "filter_byte"
    elif kind == 8 and matches:  # a match that reaches to / over its row's end
        i = matches[int(rng.integers(0, len(matches)))]
        left = stride - pos[i] % stride
        for want in (left, left + C, left - C):
            if 3 <= want <= 258 and want != T[i][1]:
                T[i] = ("match", want, T[i][2])
                name = "match_to_row_end"
                break
    elif kind == 9 and matches:  # the other distance code's bit
        i = matches[int(rng.integers(0, len(matches)))]
        T[i] = ("match", T[i][1], T[i][2] ^ 1)
        name = "dist_bit"
    elif kind == 10 and len(T) > 4:  # a token doubled or dropped
        i = int(rng.integers(0, len(T) - 1))
        if rng.random() < 0.5:
            T[i:i + 1] = [T[i], T[i]]
            name = "doubled"
        else:
            del T[i]
            name = "dropped"
    elif kind == 11 and lits:  # literal values changed (always valid: other pixels)
        for _ in range(int(rng.integers(1, 5))):
            i = lits[int(rng.integers(0, len(lits)))]
            if pos[i] % stride:
                T[i] = ("lit", int(rng.integers(0, 256)))
        name = "literals"
    return T, name


def balanced(s, T, rng):
    """makes an edited token list stand for exactly the image's bytes again (literals appended / tokens cut in front of the end-of-block
    symbol), so that the edit itself -- not the byte count -- is what the decoders judge"""
    total = s.stride * s.h
    body = [t for t in T if t[0] != "eob"]
    n = sum(1 if t[0] == "lit" else t[1] for t in body)
    while n > total and body:
        t = body.pop()
        n -= 1 if t[0] == "lit" else t[1]
    while n < total:
        col = n % s.stride
        body.append(("lit", (2 if n >= s.stride else 0) if col == 0 else int(rng.integers(0, 256))))
        n += 1
    return body + [("eob",)]


class LargeStream:
    """The same for LARGE files: the tokens come from a C loop (tests/cpp/decode_emul.cpp: fpng_emul_tokenize), edits are local and
    written by splicing bits into the stream -- megapixel images whose streams span many workgroups of the GPU decoder."""

    def __init__(self, png, plan, emul_lib):
        import ctypes as C
        res, mode, ofs, ln, first, limit, lut = plan(png)
        assert res.status == 0 and mode == 0
        self.png, self.ofs, self.ln, self.first = bytes(png), ofs, ln, first
        self.w, self.h, self.c = res.w, res.h, res.channels_in_file
        self.stride = self.w * self.c + 1
        lenof = lut[4096:].view(np.uint8)
        self.lit_code, self.len_syms, self.eob = {}, {}, None
        for i in range(4096):
            e = int(lut[i])
            L, n = entry_code_bits(e), (e >> 26) & 3
            if not L:
                continue
            if n:
                b = e & 255
                self.lit_code[b] = (i & ((1 << int(lenof[b])) - 1), int(lenof[b]))
            elif e & (1 << 25):
                self.len_syms[(e & 511, (e >> 9) & 7)] = (i & ((1 << L) - 1), L)
            else:
                self.eob = (i & ((1 << L) - 1), L)
        cap = self.stride * self.h + 16
        self.kind, self.value, self.aux, self.bitpos = np.zeros(cap, np.uint8), np.zeros(cap, np.uint16), np.zeros(cap, np.uint8), np.zeros(cap, np.uint64)
        b = np.frombuffer(self.png, dtype=np.uint8)
        emul_lib.fpng_emul_tokenize.restype = C.c_longlong
        emul_lib.fpng_emul_tokenize.argtypes = [C.c_void_p, C.c_uint32, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        n = emul_lib.fpng_emul_tokenize(b.ctypes.data, b.size, cap, self.kind.ctypes.data, self.value.ctypes.data, self.aux.ctypes.data, self.bitpos.ctypes.data)
        assert n > 0, n
        self.n = int(n)
        self.kind, self.value, self.aux, self.bitpos = self.kind[:n], self.value[:n], self.aux[:n], self.bitpos[:n]
        size = np.where(self.kind == 0, 1, np.where(self.kind == 1, self.value, 0)).astype(np.int64)
        self.out_pos = np.concatenate([[0], np.cumsum(size)[:-1]])
        self.end_bit = int(self.bitpos[-1]) + self.eob[1]
        self.zint = int.from_bytes(self.png[ofs + 8: ofs + 8 + ln], "little") & ((1 << self.end_bit) - 1)

    def token(self, i):
        k = int(self.kind[i])
        return ("lit", int(self.value[i])) if k == 0 else (("match", int(self.value[i]), int(self.aux[i])) if k == 1 else ("eob",))

    def _bits(self, tokens):
        v = n = 0
        for t in tokens:
            if t[0] == "lit":
                if t[1] not in self.lit_code:
                    return None
                code, l = self.lit_code[t[1]]
                v |= code << n
                n += l
            else:
                sym = [(base, xb) for (base, xb) in self.len_syms if base <= t[1] < base + (1 << xb)]
                if not sym:
                    return None
                base, xb = sym[0]
                code, l = self.len_syms[(base, xb)]
                v |= (code | (t[1] - base) << l | (t[2] & 1) << (l + xb)) << n
                n += l + xb + 1
        return v, n

    def splice(self, i0, i1, tokens):
        """the file with tokens [i0, i1) replaced (i1 < n: the end-of-block symbol stays); None if a token has no code in this table"""
        enc = self._bits(tokens)
        if enc is None:
            return None
        v, nb = enc
        p0, p1 = int(self.bitpos[i0]), int(self.bitpos[i1])
        bits = (self.zint & ((1 << p0) - 1)) | v << p0 | (self.zint >> p1) << (p0 + nb)
        n = p0 + nb + self.end_bit - p1
        z = bits.to_bytes((n + 7) // 8, "little") + b"\x12\x34\x56\x78"
        head = self.png[: self.ofs]
        return head + struct.pack(">I", len(z)) + b"IDAT" + z + struct.pack(">I", zlib.crc32(b"IDAT" + z)) + struct.pack(">I", 0) + b"IEND" + struct.pack(">I", zlib.crc32(b"IEND"))


def mutate_large(s, rng):
    """-> (name, file) or (name, None): ONE local edit of a LargeStream, most of them keeping the byte count (the edit itself is
    what the decoders judge), at a random place of the stream"""
    C, stride, n = s.c, s.stride, s.n
    kind = int(rng.integers(0, 10))
    i = int(rng.integers(0, n - 1))
    lo, hi = i, min(n - 1, i + 4000)
    k, v, col = s.kind[lo:hi], s.value[lo:hi].astype(np.int64), (s.out_pos[lo:hi] % stride)
    px_start = (col >= 1) & ((col - 1) % C == 0)
    lits = k == 0
    runC = lits.copy()
    for d in range(1, C):
        runC[:-d] &= lits[d:]
        runC[-d:] = False

    def pick(mask):
        idx = np.flatnonzero(mask)
        return lo + int(idx[int(rng.integers(0, len(idx)))]) if len(idx) else None
    if kind in (0, 1):  # C literals that are a pixel -> a match of one pixel; kind 1: at a row's FIRST pixel
        j = pick(px_start & runC & ((col == 1) if kind == 1 else (col > 1)))
        if j is None:
            return "none", None
        return ("lit2match_firstpx" if kind == 1 else "lit2match"), s.splice(j, j + C, [("match", C, 0)])
    if kind == 2:  # a match split in two
        j = pick((k == 1) & (v >= 2 * max(C, 3)))
        if j is None:
            return "none", None
        L = int(s.value[j])
        a = int(rng.integers(1, L // C)) * C
        if a < 3 or L - a < 3:
            return "none", None
        return "split", s.splice(j, j + 1, [("match", a, 0), ("match", L - a, 0)])
    if kind == 3:  # two neighbouring matches merged
        m = (k == 1)
        j = pick(m[:-1] & m[1:] & (v[:-1] + v[1:] <= 258)) if len(m) > 1 else None
        if j is None:
            return "none", None
        return "merge", s.splice(j, j + 2, [("match", int(s.value[j]) + int(s.value[j + 1]), 0)])
    if kind == 4:  # a match one pixel longer, the pixel of literals behind it dropped
        ok = (k[:-C] == 1) & (v[:-C] + C <= 258) & runC[1:len(k) - C + 1] if len(k) > C + 1 else np.zeros(0, bool)
        j = pick(ok) if len(ok) else None
        if j is None:
            return "none", None
        return "match_takes_next_pixel", s.splice(j, j + 1 + C, [("match", int(s.value[j]) + C, 0)])
    if kind == 5:  # a match one pixel shorter, a pixel of literals behind it instead
        j = pick((k == 1) & (v - C >= 3))
        if j is None:
            return "none", None
        return "match_gives_last_pixel", s.splice(j, j + 1, [("match", int(s.value[j]) - C, 0)] + [("lit", int(b)) for b in rng.integers(0, 256, C)])
    if kind == 6:  # literal values changed
        j = pick(lits & (col != 0))
        if j is None:
            return "none", None
        return "literal", s.splice(j, j + 1, [("lit", int(rng.integers(0, 256)))])
    if kind == 7:  # a filter literal changed
        j = pick(lits & (col == 0))
        if j is None:
            return "none", None
        return "filter_byte", s.splice(j, j + 1, [("lit", int(rng.choice([0, 1, 2, 3, 4])))])
    if kind == 8:  # a match that reaches over its row's end (a pixel of literals dropped behind it where there is one)
        j = pick(k == 1)
        if j is None:
            return "none", None
        left = stride - int(s.out_pos[j] % stride)
        want = left + C
        if not (3 <= want <= 258):
            return "none", None
        return "match_over_row_end", s.splice(j, j + 1, [("match", want, 0)])
    j = pick(k == 1)  # a match that is no whole number of pixels, the byte count made good with a literal more or less
    if j is None or j + 2 >= n or s.kind[j + 1] != 0:
        return "none", None
    L = int(s.value[j])
    if L + 1 <= 258 and rng.random() < 0.5:
        return "match_plus_a_byte", s.splice(j, j + 2, [("match", L + 1, 0)])
    if L - 1 >= 3:
        return "match_minus_a_byte", s.splice(j, j + 1, [("match", L - 1, 0), ("lit", int(rng.integers(0, 256)))])
    return "none", None
