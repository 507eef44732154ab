"""fpng token streams written again under OTHER dynamic Huffman tables (TEST INFRASTRUCTURE): random complete prefix codes with
lengths up to 12 (and beyond, to be turned away), single-code tables, HLIT / HDIST / HCLEN larger than needed, every way of spelling
the code lengths with the repeat symbols 16 / 17 / 18, distance tables of every shape the reference lets through or not
(src/fpng.cpp:1954-2105), and codes for the length symbols 286 / 287, which Deflate reserves and the reference's 4-channel decoder
takes for matches of length zero (:2668-2760: nothing happens where the previous pixel's deltas are all zero, elsewhere -- and
always in row 0 -- ONE more pixel is written).  The reference's decoder judges what such files mean."""
import struct
import zlib

import numpy as np

CLEN_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]


def complete_lengths(n, rng, maxlen):
    """n code lengths whose Kraft sum is exactly 1 (n >= 2), none longer than maxlen"""
    leaves = [1, 1]
    while len(leaves) < n:
        cand = [i for i, d in enumerate(leaves) if d < maxlen]
        i = cand[int(rng.integers(0, len(cand)))] if rng.random() < 0.5 else min(cand, key=lambda k: leaves[k])
        d = leaves.pop(i)
        leaves += [d + 1, d + 1]
    return leaves


def canonical(lengths):
    """symbol -> (code as written LSB-first, length) by the Deflate rule"""
    maxl = max(lengths) if lengths else 0
    count = [0] * (maxl + 2)
    for l in lengths:
        count[l] += 1
    count[0] = 0
    code, nxt = 0, [0] * (maxl + 2)
    for l in range(1, maxl + 1):
        code = (code + count[l - 1]) << 1
        nxt[l] = code
    out = {}
    for s, l in enumerate(lengths):
        if l:
            c = nxt[l]
            nxt[l] += 1
            out[s] = (int(format(c, f"0{l}b")[::-1], 2), l)
    return out


class BitWriter:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, bits):
        self.v |= (value & ((1 << bits) - 1)) << self.n
        self.n += bits

    def bytes(self):
        return self.v.to_bytes((self.n + 7) // 8, "little")


def spell_lengths(lens, rng):
    """[(code length symbol, extra value, extra bits)] for a list of code lengths, repeats used at random"""
    out, i = [], 0
    while i < len(lens):
        v = lens[i]
        run = 1
        while i + run < len(lens) and lens[i + run] == v:
            run += 1
        if v == 0 and run >= 3 and rng.random() < 0.8:
            if run >= 11 and rng.random() < 0.7:
                r = min(run, 138) if rng.random() < 0.7 else int(rng.integers(11, min(run, 138) + 1))
                out.append((18, r - 11, 7))
            else:
                r = min(run, 10) if rng.random() < 0.7 else int(rng.integers(3, min(run, 10) + 1))
                out.append((17, r - 3, 3))
            i += r
        elif v != 0 and i > 0 and lens[i - 1] == v and run >= 3 and rng.random() < 0.8:
            r = min(run, 6) if rng.random() < 0.7 else int(rng.integers(3, min(run, 6) + 1))
            out.append((16, r - 3, 2))
            i += r
        else:
            out.append((v, 0, 0))
            i += 1
    return out


def reencode(stream, rng, tokens=None):
    """-> (what was done, file): stream.tokens (tests/token_mutator.py) or `tokens` under a new random table.  Tokens may also be
    ('sym', 286 or 287): a reserved length symbol followed by the distance bit."""
    T = list(tokens if tokens is not None else stream.tokens)
    C = stream.c
    notes = []
    used = set()
    for t in T:
        if t[0] == "lit":
            used.add(t[1])
        elif t[0] == "match":
            k = max(i for i in range(29) if LEN_BASE[i] <= t[1])
            used.add(257 + k)
        elif t[0] == "sym":
            used.add(t[1])
        else:
            used.add(256)
    used.add(256)
    extra = int(rng.integers(0, 4))
    if extra == 1:  # more symbols get codes than the stream uses
        for s in rng.integers(0, 286, int(rng.integers(1, 120))):
            used.add(int(s))
    elif extra == 2:  # every symbol
        used |= set(range(286))
        notes.append("all_symbols")
    if rng.random() < 0.15:
        used |= {286} if rng.random() < 0.5 else {286, 287}
        notes.append("reserved_symbols_coded")
    syms = sorted(used)
    maxlen = 12
    if rng.random() < 0.06:
        maxlen = int(rng.integers(13, 16))
        notes.append("codes_longer_than_12")
    if len(syms) == 1:
        lens_used = [int(rng.integers(1, 13))]
        notes.append("single_code")
    else:
        lens_used = complete_lengths(len(syms), rng, maxlen)
        if maxlen > 12 and max(lens_used) <= 12:
            notes.remove("codes_longer_than_12")
    perm = rng.permutation(len(syms))
    lit_lens = [0] * 288
    for k, s in enumerate(syms):
        lit_lens[s] = lens_used[int(perm[k])]
    if rng.random() < 0.05 and len(syms) > 2:  # an incomplete / oversubscribed code
        s = syms[int(rng.integers(0, len(syms)))]
        lit_lens[s] = max(1, min(12, lit_lens[s] + int(rng.choice([-1, 1]))))
        notes.append("kraft_off")
    n_lit = max(257, max(s for s in range(288) if lit_lens[s]) + 1)
    if rng.random() < 0.4:
        n_lit = int(rng.integers(n_lit, 289))
    # distance code lengths: the reference wants one or two 1-bit codes, symbol C - 1 among them (and C, if two)
    shape = int(rng.integers(0, 8))
    n_dist = int(rng.integers(C + 1, 33)) if rng.random() < 0.5 else C + 1
    dist = [0] * n_dist
    if shape <= 2:
        dist[C - 1] = 1
        notes.append("one_dist_code")
    elif shape <= 4:
        dist[C - 1] = dist[C] = 1
        notes.append("two_dist_codes")
    elif shape == 5:
        dist[C - 1] = 1
        for k in rng.integers(0, n_dist, 3):
            if dist[int(k)] == 0:
                dist[int(k)] = int(rng.integers(2, 8))
        notes.append("other_dist_lengths")
    elif shape == 6:
        dist[int(rng.integers(0, n_dist))] = 1
        if rng.random() < 0.5:
            dist[int(rng.integers(0, n_dist))] = 1
        notes.append("dist_code_anywhere")
    else:
        k = int(rng.integers(0, 4))
        for j in range(k):
            dist[(C - 1 + j) % n_dist] = 1
        notes.append(f"{k}_dist_codes")
    if rng.random() < 0.3:
        n_dist = max(C + 1, max([i for i in range(n_dist) if dist[i]] + [0]) + 1)
        dist = dist[:n_dist]
    all_lens = lit_lens[:n_lit] + dist
    spelled = spell_lengths(all_lens, rng)
    cl_used = sorted({s for s, _, _ in spelled})
    cl_lens = [0] * 19
    if len(cl_used) == 1:
        cl_lens[cl_used[0]] = int(rng.integers(1, 8))
    else:
        ls = complete_lengths(len(cl_used), rng, 7)
        p2 = rng.permutation(len(cl_used))
        for k, s in enumerate(cl_used):
            cl_lens[s] = ls[int(p2[k])]
    n_clen = max(4, max(i for i in range(19) if cl_lens[CLEN_ORDER[i]]) + 1)
    if rng.random() < 0.3:
        n_clen = int(rng.integers(n_clen, 20))
    cl_code = canonical(cl_lens)
    w = BitWriter()
    w.put(0x78, 8), w.put(0x01, 8)
    w.put(1, 1), w.put(2, 2)
    w.put(n_lit - 257, 5), w.put(n_dist - 1, 5), w.put(n_clen - 4, 4)
    for i in range(n_clen):
        w.put(cl_lens[CLEN_ORDER[i]], 3)
    for s, ev, eb in spelled:
        w.put(*cl_code[s])
        if eb:
            w.put(ev, eb)
    code = canonical(lit_lens)
    for t in T:
        if t[0] == "lit":
            w.put(*code[t[1]])
        elif t[0] == "match":
            k = max(i for i in range(29) if LEN_BASE[i] <= t[1])
            w.put(*code[257 + k])
            if LEN_EXTRA[k]:
                w.put(t[1] - LEN_BASE[k], LEN_EXTRA[k])
            w.put(t[2] & 1, 1)
        elif t[0] == "sym":
            w.put(*code[t[1]])
            w.put(int(rng.integers(0, 2)), 1)
        else:
            w.put(*code[256])
    z = w.bytes() + b"\x12\x34\x56\x78"
    head = stream.png[: stream.ofs]
    f = head + struct.pack(">I", len(z)) + b"IDAT" + z + struct.pack(">I", zlib.crc32(b"IDAT" + z)) + struct.pack(">I", 0) + b"IEND" + struct.pack(">I", zlib.crc32(b"IEND"))
    return "+".join(notes) or "plain", f


def with_reserved_symbols(stream, rng):
    """a copy of the stream's tokens with a few ('sym', 286 / 287) tokens: in front of a pixel's literals, or in their place (where
    the reference writes a pixel for the symbol the row keeps its length), now and then anywhere"""
    T = list(stream.tokens)
    C = stream.c
    for _ in range(int(rng.integers(1, 4))):
        pos = stream.positions(T)
        starts = [i for i, t in enumerate(T) if t[0] == "lit" and (pos[i] % stream.stride) >= 1 and (pos[i] % stream.stride - 1) % C == 0 and
                  i + C <= len(T) and all(T[i + k][0] == "lit" for k in range(C))]
        sym = ("sym", 286 + int(rng.integers(0, 2)))
        if starts and rng.random() < 0.85:
            i = starts[int(rng.integers(0, len(starts)))]
            if rng.random() < 0.5:
                T[i:i + C] = [sym]
            else:
                T.insert(i, sym)
        else:
            T.insert(int(rng.integers(0, len(T))), sym)
    return T
