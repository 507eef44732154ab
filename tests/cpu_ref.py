"""ctypes doors into the CPU checkers (TEST INFRASTRUCTURE).

`oracle`  : oracle/libfpng_oracle.so  -- the row-local C restatement (always available; built by
            __graft_entry__.build() / `make -C oracle`).
`ref`     : oracle/_ref/libfpng_ref.so -- the UNMODIFIED reference compiled from /root/reference
            (present in the dev container and, as a prebuilt file, on the GPU box).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def _ensure_built():
    so = os.path.join(ORACLE_DIR, "libfpng_oracle.so")
    src = os.path.join(ORACLE_DIR, "fpng_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


class _Oracle:
    def __init__(self):
        _ensure_built()
        L = C.CDLL(os.path.join(ORACLE_DIR, "libfpng_oracle.so"))
        L.fpo_crc32.restype = C.c_uint32
        L.fpo_crc32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.fpo_adler32.restype = C.c_uint32
        L.fpo_adler32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.fpo_max_encoded_size.restype = C.c_size_t
        L.fpo_max_encoded_size.argtypes = [C.c_uint32] * 3
        L.fpo_encode.restype = C.c_int
        L.fpo_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                 C.POINTER(C.c_size_t)]
        L.fpo_encode_band_1pass.restype = C.c_uint64
        L.fpo_encode_band_1pass.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.fpo_get_1pass_table.restype = None
        L.fpo_get_1pass_table.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.fpo_get_len_tables.restype = None
        L.fpo_get_len_tables.argtypes = [C.c_void_p, C.c_void_p]
        L.fpo_build_dynamic_table.restype = C.c_uint32
        L.fpo_build_dynamic_table.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.fpo_train_tables.restype = C.c_uint32
        L.fpo_train_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L = L

    def crc32(self, data, prev=0):
        b = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        return self.L.fpo_crc32(b.ctypes.data, b.size, prev)

    def adler32(self, data, prev=1):
        b = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        return self.L.fpo_adler32(b.ctypes.data, b.size, prev)

    def max_size(self, w, h, c):
        return self.L.fpo_max_encoded_size(w, h, c)

    def encode(self, img, w, h, c, flags=0):
        """img: uint8 array of w*h*c bytes -> PNG bytes (or None on bad args)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        cap = self.max_size(w, h, c) if (w and h and c in (3, 4)) else 64
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        ok = self.L.fpo_encode(img.ctypes.data, w, h, c, flags, out.ctypes.data, cap, C.byref(n))
        return out[: n.value].tobytes() if ok else None

    def band_1pass(self, img, w, h, c, y0, y1):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        cap = ((w * c + 1) * (y1 - y0) * 12 + 7) // 8 + 64
        out = np.zeros(cap, dtype=np.uint8)
        s1, s2, ln, lu = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0), C.c_uint32(0)
        bits = self.L.fpo_encode_band_1pass(img.ctypes.data, w, h, c, y0, y1, out.ctypes.data, cap, C.byref(s1),
                                            C.byref(s2), C.byref(ln), C.byref(lu))
        self.last_unit_bits = lu.value
        return bits, out[: (bits + 7) // 8].copy(), s1.value, s2.value, ln.value

    def table_1pass(self, c):
        lens = np.zeros(288, dtype=np.uint8)
        codes = np.zeros(288, dtype=np.uint16)
        p = C.c_void_p()
        plen, sbit = C.c_uint32(0), C.c_uint32(0)
        self.L.fpo_get_1pass_table(c, lens.ctypes.data, codes.ctypes.data, C.byref(p), C.byref(plen), C.byref(sbit))
        prefix = C.string_at(p, plen.value)
        return lens, codes, prefix, sbit.value

    def len_tables(self):
        sym = np.zeros(256, dtype=np.uint16)
        ex = np.zeros(256, dtype=np.uint8)
        self.L.fpo_get_len_tables(sym.ctypes.data, ex.ctypes.data)
        return sym, ex

    def build_dynamic_table(self, hist, c):
        hist = np.ascontiguousarray(hist, dtype=np.uint32)
        lens = np.zeros(288, dtype=np.uint8)
        codes = np.zeros(288, dtype=np.uint16)
        hdr = np.zeros(400, dtype=np.uint8)
        bits = self.L.fpo_build_dynamic_table(hist.ctypes.data, c, lens.ctypes.data, codes.ctypes.data, hdr.ctypes.data)
        return lens, codes, hdr, bits


    def train_tables(self, imgs, c):
        """-> dict like tests/golden/train.json: prefix bytes (hex), pending bits, codes, code sizes."""
        imgs = [np.ascontiguousarray(i, dtype=np.uint8) for i in imgs]
        n = len(imgs)
        ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        ws = (C.c_uint32 * n)(*[i.shape[1] for i in imgs])
        hs = (C.c_uint32 * n)(*[i.shape[0] for i in imgs])
        lens = np.zeros(288, dtype=np.uint8)
        codes = np.zeros(288, dtype=np.uint16)
        hdr = np.zeros(400, dtype=np.uint8)
        bits = self.L.fpo_train_tables(ptrs, ws, hs, n, c, lens.ctypes.data, codes.ctypes.data, hdr.ctypes.data)
        return table_record(hdr, bits, codes, lens)


def table_record(hdr, bits, codes, lens):
    """Header bytes + bit count of a block prefix in the form the reference's training mode prints it (src/fpng_test.cpp:905-925):
    whole bytes, then the pending bits."""
    nb = bits // 8
    return {"prefix": bytes(hdr[:nb]).hex(), "bit_buf": int(hdr[nb]) & ((1 << (bits % 8)) - 1), "bit_buf_size": bits % 8,
            "codes": [int(v) for v in codes], "code_sizes": [int(v) for v in lens]}


class _Ref:
    def __init__(self):
        path = os.path.join(ORACLE_DIR, "_ref", "libfpng_ref.so")
        if not os.path.exists(path):
            _ensure_built()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = C.CDLL(path)
        L.ref_init.restype = None
        L.ref_supports_sse41.restype = C.c_int
        L.ref_crc32.restype = C.c_uint32
        L.ref_crc32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.ref_adler32.restype = C.c_uint32
        L.ref_adler32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        L.ref_encode.restype = C.c_int
        L.ref_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t,
                                 C.POINTER(C.c_size_t)]
        L.ref_get_info.restype = C.c_int
        L.ref_get_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32)]
        L.ref_decode.restype = C.c_int
        L.ref_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32]
        L.ref_time_encode.restype = C.c_double
        L.ref_time_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                      C.POINTER(C.c_size_t)]
        if hasattr(L, "ref_time_decode"):
            L.ref_time_decode.restype = C.c_double
            L.ref_time_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.ref_init()
        self.L = L

    def crc32(self, data, prev=0):
        b = np.ascontiguousarray(data, dtype=np.uint8)
        return self.L.ref_crc32(b.ctypes.data, b.size, prev)

    def adler32(self, data, prev=1):
        b = np.ascontiguousarray(data, dtype=np.uint8)
        return self.L.ref_adler32(b.ctypes.data, b.size, prev)

    def encode(self, img, w, h, c, flags=0):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        n_f = (w * c + 1) * h
        cap = 58 + 6 + n_f + 5 * ((n_f + 65534) // 65535) + 16 + 64
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        ok = self.L.ref_encode(img.ctypes.data, w, h, c, flags, out.ctypes.data, cap, C.byref(n))
        return out[: n.value].tobytes() if ok else None

    def decode(self, png, desired):
        b = np.frombuffer(png, dtype=np.uint8)
        w, h, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        st = self.L.ref_get_info(b.ctypes.data, b.size, C.byref(w), C.byref(h), C.byref(c))
        if st != 0:
            return st, None, 0, 0, 0
        out = np.zeros(w.value * h.value * desired, dtype=np.uint8)
        st = self.L.ref_decode(b.ctypes.data, b.size, out.ctypes.data, out.size, C.byref(w), C.byref(h), C.byref(c),
                               desired)
        return st, out, w.value, h.value, c.value

    def time_decode(self, png, desired, reps=3):
        """best seconds of `reps` fpng_decode_memory() calls into one reused vector (the reference harness's way)"""
        b = np.frombuffer(png, dtype=np.uint8)
        secs = self.L.ref_time_decode(b.ctypes.data, b.size, desired, reps)
        assert secs > 0, f"reference decoder status {-secs}"
        return secs

    def time_encode(self, img, w, h, c, flags=0, reps=3):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        n = C.c_size_t(0)
        return self.L.ref_time_encode(img.ctypes.data, w, h, c, flags, reps, C.byref(n)), n.value


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = _Oracle()
    return _oracle


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libfpng_ref.so")) or os.path.exists(
        "/root/reference/src/fpng.cpp")


def ref():
    global _ref
    if _ref is None:
        _ref = _Ref()
    return _ref


# ---------------------------------------------------------------------------------------------
# Fuzz image recipe that actually reaches the edge cases
# ---------------------------------------------------------------------------------------------
def fuzz_image(rng, force_dims=None, c=None):
    """Returns (img uint8[h,w,c], w, h, c): random runs / vertical copies / deltas / noise."""
    if c is None:
        c = int(rng.integers(3, 5))
    if force_dims is not None:
        w, h = force_dims
    else:
        u = rng.random()
        if u < 0.15:
            w, h = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        elif u < 0.30:
            w, h = int(rng.integers(60, 401)), int(rng.integers(1, 5))
        else:
            w, h = int(rng.integers(1, 49)), int(rng.integers(1, 15))
    q = rng.random() ** 2
    p_run = rng.random() * 0.7
    p_up = rng.random() * 0.5
    amp = int(rng.choice([1, 2, 4, 16, 256]))
    img = np.zeros((h, w, c), dtype=np.uint8)
    r = rng.random((h, w, 3))
    noise = rng.integers(0, 256, size=(h, w, c), dtype=np.int64)
    delta = rng.integers(0, amp, size=(h, w, c), dtype=np.int64) if amp > 1 else np.zeros((h, w, c), dtype=np.int64)
    for y in range(h):
        for x in range(w):
            if r[y, x, 0] < q:
                img[y, x] = noise[y, x]
            elif x > 0 and r[y, x, 1] < p_run:
                # copy-left in FILTERED space: pixel - up == left - left_up
                if y > 0:
                    img[y, x] = (img[y, x - 1].astype(np.int64) - img[y - 1, x - 1] + img[y - 1, x]) & 0xFF
                else:
                    img[y, x] = img[y, x - 1]
            elif y > 0 and r[y, x, 2] < p_up:
                img[y, x] = img[y - 1, x]
            else:
                base = img[y - 1, x].astype(np.int64) if y > 0 else np.zeros(c, dtype=np.int64)
                img[y, x] = (base + delta[y, x]) & 0xFF
    return img, w, h, c
