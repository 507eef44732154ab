"""Chunk-level and block-level edits of fpng files with their CRCs made good again (TEST INFRASTRUCTURE).  A flipped bit in a PNG
container nearly always ends at the chunk's CRC; these edits reach what lies behind it: the container walk of the reference
(src/fpng.cpp:2930-3077: chunk order, the fdEC marker, IHDR fields, dimensions, ancillary / unknown critical chunks, several
IDATs, what follows IEND), the zlib header, the stored-block layout (:2107-2207) and the dynamic block header (:1954-2105).
The reference's fpng_get_info / fpng_decode_memory are the judges of what such a file means."""
import struct
import zlib

SIG = b"\x89PNG\r\n\x1a\n"


def chunk(t, body, bad_crc=False):
    crc = zlib.crc32(t + body) ^ (0x5A5A if bad_crc else 0)
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", crc & 0xFFFFFFFF)


def chunks_of(png):
    """[(type, body)] of a well-formed file"""
    out, o = [], 8
    while o + 12 <= len(png):
        n = struct.unpack(">I", png[o:o + 4])[0]
        out.append((bytes(png[o + 4:o + 8]), bytes(png[o + 8:o + 8 + n])))
        o += 12 + n
    return out


def build(chunks, tail=b""):
    return SIG + b"".join(chunk(t, b) if len(x) == 0 else chunk(t, b, True) for (t, b, *x) in chunks) + tail


def stored_raw(body):
    """the filtered bytes a stored-block zlib stream holds"""
    raw, o = bytearray(), 2
    while o + 5 <= len(body) - 4:
        n = body[o + 1] | body[o + 2] << 8
        raw += body[o + 5:o + 5 + n]
        final = body[o] & 1
        o += 5 + n
        if final:
            break
    return bytes(raw)


def pack_stored(body, raw, sizes):
    """`raw` as stored blocks of the given sizes (the last size repeats) between the zlib header and the Adler-32 of `body`"""
    out, p, k = bytearray(body[:2]), 0, 0
    sizes = [min(int(v), 65535) for v in sizes]
    while True:
        n = min(sizes[min(k, len(sizes) - 1)], len(raw) - p)
        last = p + n >= len(raw)
        out += bytes([1 if last else 0]) + struct.pack("<HH", n, n ^ 0xFFFF) + raw[p:p + n]
        p += n
        k += 1
        if last:
            break
    return bytes(out) + body[-4:]


def restore_stored(body, sizes):
    """a stored-block zlib stream cut into blocks of other sizes"""
    return pack_stored(body, stored_raw(body), sizes)


def stored_with_tail(png, tail, sizes=(65535,)):
    """a stored-block file with `tail` appended to its filtered bytes (the reference lets exactly ONE zero byte pass: it takes it
    for the filter byte of a row that never comes, src/fpng.cpp:2158-2166)"""
    ch = chunks_of(png)
    i = [t for t, _ in ch].index(b"IDAT")
    body = ch[i][1]
    assert (body[2] & 6) == 0
    ch[i] = (b"IDAT", pack_stored(body, stored_raw(body) + bytes(tail), sizes))
    return build(ch)


def mutate(png, rng):
    """-> (name, edited file)"""
    ch = chunks_of(png)
    types = [t for t, _ in ch]
    ihdr = bytearray(ch[0][1])
    idat_i = types.index(b"IDAT")
    fdec_i = types.index(b"fdEC")
    kind = int(rng.integers(0, 30))
    tail = b""
    name = f"k{kind}"
    anc = (b"tEXt", b"Comment\0made by a test") if rng.random() < 0.5 else (b"gAMA", struct.pack(">I", 45455))
    if kind == 0:
        ch.insert(int(rng.integers(1, len(ch))), anc); name = "ancillary_inserted"
    elif kind == 1:
        ch.insert(int(rng.integers(1, len(ch))), (b"ABCD", b"critical")); name = "unknown_critical"
    elif kind == 2:
        del ch[fdec_i]; name = "no_fdEC"
    elif kind == 3:
        ch.insert(fdec_i, ch[fdec_i]); name = "two_fdEC"
    elif kind == 4:
        body = bytearray(ch[fdec_i][1])
        if rng.random() < 0.5:
            body[int(rng.integers(0, len(body)))] ^= 1 << int(rng.integers(0, 8))
        else:
            body = body[: int(rng.integers(0, len(body)))] if rng.random() < 0.5 else body + b"\0"
        ch[fdec_i] = (b"fdEC", bytes(body)); name = "fdEC_payload"
    elif kind == 5:
        f = ch.pop(fdec_i); ch.insert(types.index(b"IDAT"), f); name = "fdEC_behind_IDAT"  # (index of IDAT before the pop = one behind it now)
    elif kind == 6:
        b = ch[idat_i][1]; cut = int(rng.integers(1, max(2, len(b)))); ch[idat_i:idat_i + 1] = [(b"IDAT", b[:cut]), (b"IDAT", b[cut:])]; name = "two_IDAT"
    elif kind == 7:
        ch[idat_i] = (b"IDAT", ch[idat_i][1][: int(rng.integers(0, 8))]); name = "short_IDAT"
    elif kind == 8:
        ihdr[8] = int(rng.choice([1, 2, 4, 16])); ch[0] = (b"IHDR", bytes(ihdr)); name = "bit_depth"
    elif kind == 9:
        ihdr[9] = int(rng.choice([0, 3, 4, 2 if ihdr[9] == 6 else 6])); ch[0] = (b"IHDR", bytes(ihdr)); name = "colour_type"
    elif kind == 10:
        ihdr[10 + int(rng.integers(0, 3))] = int(rng.integers(1, 256)); ch[0] = (b"IHDR", bytes(ihdr)); name = "method_fields"
    elif kind == 11:
        w, h = struct.unpack(">II", ihdr[:8])
        w2, h2 = [(0, h), (w, 0), (w + 1, h), (w, h + 1), (max(1, w - 1), h), (w, max(1, h - 1)), ((1 << 24) + 1, 1), (1 << 24, 65), (40000, 40000), (0x80000000, 1), (h, w)][int(rng.integers(0, 11))]
        ihdr[:8] = struct.pack(">II", w2, h2); ch[0] = (b"IHDR", bytes(ihdr)); name = "dimensions"
    elif kind == 12:
        del ch[-1]; name = "no_IEND"
    elif kind == 13:
        tail = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype="uint8")); name = "junk_behind_IEND"
    elif kind == 14:
        ch[0] = (b"IHDR", bytes(ihdr) + b"\0"); name = "IHDR_length"
    elif kind == 15:
        i = int(rng.integers(0, len(ch))); ch[i] = (ch[i][0], ch[i][1], "badcrc"); name = "bad_crc_" + ch[i][0].decode()
    elif kind == 16:
        ch.insert(int(rng.integers(1, len(ch))), (b"tE1t", b"x")); name = "chunk_type_not_letters"
    elif kind == 17:
        ch[-1] = (b"IEND", b"\0\0"); name = "IEND_with_body"
    elif kind == 18:
        ch.insert(1, anc); ch.insert(len(ch) - 1, anc); name = "ancillary_twice"
    elif kind == 19:  # zlib header bytes
        b = bytearray(ch[idat_i][1]); b[int(rng.integers(0, 2))] = int(rng.choice([0x78, 0x01, 0x9C, 0xDA, 0x08, 0x5E])); ch[idat_i] = (b"IDAT", bytes(b)); name = "zlib_header"
    elif kind == 20:  # the block's type bits
        b = bytearray(ch[idat_i][1]); b[2] ^= int(rng.choice([1, 2, 4, 6, 7])); ch[idat_i] = (b"IDAT", bytes(b)); name = "block_type_bits"
    elif kind == 21:  # a chunk whose length reaches over the end of the file
        out = build(ch)
        o = 8 + 25 + (12 + len(ch[1][1]) if len(ch) > 1 else 0)
        out = out[:o] + struct.pack(">I", 0x7FFFFFFF if rng.random() < 0.5 else len(out)) + out[o + 4:]
        return "chunk_length_over_the_end", out
    elif kind == 22:  # the fdEC marker in front of IHDR's place / IHDR missing
        if rng.random() < 0.5:
            del ch[0]; name = "no_IHDR"
        else:
            ch[0], ch[1] = ch[1], ch[0]; name = "IHDR_second"
    elif kind in (23, 24, 25):  # stored-block files: other block sizes, damaged block headers (only files that ARE stored blocks)
        b = ch[idat_i][1]
        if len(b) > 7 and (b[2] & 6) == 0:
            if kind == 23:
                sizes = [int(rng.integers(1, 70)), int(rng.integers(1, 3000)), 65535][int(rng.integers(0, 3)):][:2] if rng.random() < 0.7 else [int(rng.integers(1, 400))]
                ch[idat_i] = (b"IDAT", restore_stored(b, sizes)); name = "stored_other_block_sizes"
            elif kind == 24:
                bb = bytearray(b); bb[3 + int(rng.integers(0, 4))] ^= 1 << int(rng.integers(0, 8)); ch[idat_i] = (b"IDAT", bytes(bb)); name = "stored_len_nlen"
            else:
                bb = bytearray(restore_stored(b, [int(rng.integers(8, 200))]))
                # a filter byte that is not 0 somewhere, or the final flag too early / missing
                if rng.random() < 0.5:
                    bb[2] |= 1; name = "stored_final_flag_early"
                else:
                    bb[-5 - int(rng.integers(0, min(60, len(bb) - 12)))] ^= 0x40; name = "stored_payload_bit"
                ch[idat_i] = (b"IDAT", bytes(bb))
        else:
            name = "none"
    elif kind in (26, 27):  # dynamic header: HLIT / HDIST / HCLEN and the code length code's lengths
        b = bytearray(ch[idat_i][1])
        if len(b) > 12 and (b[2] & 6) == 4:
            v = int.from_bytes(b[2:12], "little")
            if kind == 26:
                field, width = [(3, 5), (8, 5), (13, 4)][int(rng.integers(0, 3))]
                v ^= int(rng.integers(1, 1 << width)) << field; name = "dynamic_counts"
            else:
                v ^= 1 << int(rng.integers(17, 74)); name = "dynamic_code_length_codes"
            b[2:12] = v.to_bytes(10, "little"); ch[idat_i] = (b"IDAT", bytes(b))
        else:
            name = "none"
    elif kind == 28:  # IDAT emptied of everything but the zlib header and an Adler-32
        ch[idat_i] = (b"IDAT", ch[idat_i][1][:2] + b"\x03\x00" + b"\0\0\0\1"); name = "empty_fixed_block"
    else:  # the IDAT of another shape: cut short in front of its Adler-32 / bytes appended
        b = ch[idat_i][1]
        ch[idat_i] = (b"IDAT", b[:-4] if rng.random() < 0.5 else b + bytes(int(rng.integers(1, 9)))); name = "IDAT_tail"
    return name, build(ch, tail)
