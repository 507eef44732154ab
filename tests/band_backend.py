"""CPU stand-in for the per-band GPU work, built on the oracle (TEST INFRASTRUCTURE).  It lets the
world_size-2 gloo tests exercise the real orchestration of fpng_amd/sharded.py (record all_gather,
start-bit prefix sums, Adler combine, failure rule, seam OR-merge, wrap) without a GPU."""
import zlib

import numpy as np
import torch

from cpu_ref import oracle
from fpng_amd.sharded import BandStats


class OracleBandBackend:
    def __init__(self, image):
        self.image = np.ascontiguousarray(image)  # the whole image (the oracle filters from it directly)
        self.h, self.w, self.c = self.image.shape

    def layout(self, c):
        lens, codes, prefix, sbit = oracle().table_1pass(c)
        return sbit, int(lens[256]), len(prefix)

    def count(self, rows, row_above, w, c, y0, y1):
        bits, buf, s1, s2, ln = oracle().band_1pass(self.image, w, self.h, c, y0, y1)
        return BandStats(bits, s1, s2, ln, oracle().last_unit_bits)

    def emit(self, rows, row_above, w, c, y0, y1, start_bit, is_first, is_last, adler):
        lens, codes, prefix, sbit = oracle().table_1pass(c)
        bits, buf, *_ = oracle().band_1pass(self.image, w, self.h, c, y0, y1)
        val = int.from_bytes(buf.tobytes(), "little")
        first_byte = 0 if is_first else start_bit >> 3
        acc = val << (start_bit - 8 * first_byte)
        end = start_bit + bits
        if is_first:
            tail_bits = sbit - 8 * len(prefix)
            tail_val = {3: 30, 4: 1}[c]
            assert start_bit == sbit
            acc |= int.from_bytes(prefix, "little") | (tail_val << (8 * len(prefix)))
            assert tail_bits == {3: 7, 4: 2}[c]
        if is_last:
            acc |= int(codes[256]) << (end - 8 * first_byte)
            end += int(lens[256])
            end = (end + 7) & ~7
            acc |= int.from_bytes(adler.to_bytes(4, "big"), "little") << (end - 8 * first_byte)
            end += 32
        nbytes = ((end + 7) >> 3) - first_byte
        return torch.from_numpy(np.frombuffer(acc.to_bytes(nbytes, "little"), dtype=np.uint8).copy())

    def wrap(self, png_buf, zlib_size, w, h, c):
        z = bytes(png_buf[58:58 + zlib_size].numpy())
        whole = oracle().encode(self.image, w, h, c, 0)  # container bytes (header) from the oracle
        hdr = bytearray(whole[:58])
        hdr[50:54] = zlib_size.to_bytes(4, "big")
        out = bytes(hdr) + z + zlib.crc32(b"IDAT" + z).to_bytes(4, "big") + whole[-12:]
        return torch.from_numpy(np.frombuffer(out, dtype=np.uint8).copy())

    def encode_whole(self, image, w, h, c, flags):
        return oracle().encode(np.asarray(image), w, h, c, flags)
