"""CPU stand-in for the per-band GPU work, built on the oracle (TEST INFRASTRUCTURE).  It lets the
world_size-2 gloo tests exercise the real orchestration of fpng_amd/sharded.py (histogram all_reduce,
record all_gather, start-bit prefix sums, Adler combine, failure rule, window merge, wrap) without a GPU."""
import ctypes as C
import zlib

import numpy as np
import torch

from cpu_ref import oracle
from fpng_amd.sharded import BandStats


class OracleBandBackend:
    # The stand-in shards the IDAT CRC like the GPU backend does, in its own geometry: the LINEAR part of CRC-32,
    # lin(M) = crc32(M) ^ crc32(zeros(len M)), of the band's window laid over an otherwise zero stream goes into
    # partial 0 (the other entries stay 0); XOR-ed over the bands that must be lin(whole stream), which wrap() checks.
    has_crc_partials = True

    def __init__(self, image):
        self.image = np.ascontiguousarray(image)  # the whole image (the oracle filters from it directly)
        self.h, self.w, self.c = self.image.shape
        L = oracle().L
        L.fpo_band_hist.restype = None
        L.fpo_band_hist.argtypes = [C.c_void_p] + [C.c_uint32] * 5 + [C.c_void_p]
        L.fpo_encode_band.restype = C.c_uint64
        L.fpo_encode_band.argtypes = [C.c_void_p] + [C.c_uint32] * 5 + [C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 8
        self.L = L

    def hist(self, rows, row_above, w, c, y0, y1, h):
        hist = np.zeros(288, dtype=np.uint32)
        self.L.fpo_band_hist(self.image.ctypes.data, w, h, c, y0, y1, hist.ctypes.data)
        return torch.from_numpy(hist.astype(np.int32))

    def encode(self, rows, row_above, w, c, y0, y1, h, flags, hist):
        cap = ((w * c + 1) * (y1 - y0) * 12 + 7) // 8 + 64
        out = np.zeros(cap, dtype=np.uint8)
        s1, s2, lu, ftb, eobb, eobc = (C.c_uint32(0) for _ in range(6))
        ln = C.c_uint64(0)
        hdr = np.zeros(400, dtype=np.uint8)
        hp = None
        if flags & 1:
            self._hist = np.ascontiguousarray(hist.numpy().astype(np.uint32))
            hp = self._hist.ctypes.data
        bits = self.L.fpo_encode_band(self.image.ctypes.data, w, h, c, y0, y1, hp, out.ctypes.data, cap, C.byref(s1), C.byref(s2),
                                      C.byref(ln), C.byref(lu), C.byref(ftb), C.byref(eobb), C.byref(eobc), hdr.ctypes.data)
        self._band = dict(bits=int(bits), val=int.from_bytes(out[: (bits + 7) // 8].tobytes(), "little"), first=y0 == 0, last=y1 == h,
                          ftb=ftb.value, eob_bits=eobb.value, eob_code=eobc.value, hdr=hdr)
        return BandStats(int(bits), s1.value, s2.value, ln.value, lu.value, ftb.value, eobb.value)

    def place(self, start_bit, zlib_size, token_bits, device, out=None):
        off, win = self._place(start_bit, zlib_size, token_bits, device)
        self._placed = (off, win.clone(), zlib_size)
        if out is not None:
            out[:win.numel()] = win
            win = out[:win.numel()]
        return off, win

    def _place(self, start_bit, zlib_size, token_bits, device):
        b = self._band
        fb0 = 58 * 8 + start_bit
        wb0 = 0 if b["first"] else (fb0 >> 3) & ~15
        acc = b["val"] << (fb0 - 8 * wb0)
        end = fb0 + b["bits"]
        if b["first"]:
            assert start_bit == b["ftb"]
            acc |= int.from_bytes(b["hdr"][: (b["ftb"] + 7) // 8].tobytes(), "little") << (58 * 8)
        if b["last"]:
            acc |= b["eob_code"] << (end - 8 * wb0)
            end += b["eob_bits"]
        wb1 = (((end + 7) >> 3) + 15) & ~15
        win = np.frombuffer(acc.to_bytes(wb1 - wb0, "little"), dtype=np.uint8).copy()
        if b["first"]:
            win[:58] = 0xEE  # undefined bytes: the merge must not use them
        return wb0, torch.from_numpy(win)

    @staticmethod
    def _lin_crc(data):
        return zlib.crc32(data) ^ zlib.crc32(bytes(len(data)))

    def crc_partials(self, device):
        off, win, zlib_size = self._placed
        n_data = zlib_size - 4
        buf = np.zeros(n_data, dtype=np.uint8)  # the stream without its Adler-32, zero where other bands' bits are
        w = win.numpy()
        lo, hi = max(off, 58), min(off + len(w), 58 + n_data)
        if hi > lo:
            buf[lo - 58:hi - 58] = w[lo - off:hi - off]
        n_part = ((((58 + zlib_size - 4) + 15) & ~15) - 48 + 65535) >> 16
        part = np.zeros(n_part, dtype=np.uint32)
        part[0] = self._lin_crc(buf.tobytes())
        return torch.from_numpy(part.view(np.int32).copy())

    def wrap(self, png_buf, zlib_size, adler, w, h, c, crc_partials=None):
        if crc_partials is not None:  # the bands' shares, XOR-ed by the orchestration
            got = int(crc_partials.numpy().view(np.uint32)[0])
            assert got == self._lin_crc(bytes(png_buf[58:58 + zlib_size - 4].numpy())), "sharded CRC does not add up"
            assert not crc_partials.numpy()[1:].any()
        z = bytes(png_buf[58:58 + zlib_size - 4].numpy()) + adler.to_bytes(4, "big")
        whole = oracle().encode(self.image, w, h, c, 0)  # container bytes (header) from the oracle
        hdr = bytearray(whole[:58])
        hdr[50:54] = zlib_size.to_bytes(4, "big")
        out = bytes(hdr) + z + zlib.crc32(b"IDAT" + z).to_bytes(4, "big") + whole[-12:]
        return torch.from_numpy(np.frombuffer(out, dtype=np.uint8).copy())

    def encode_whole(self, image, w, h, c, flags):
        return oracle().encode(np.asarray(image), w, h, c, flags)
