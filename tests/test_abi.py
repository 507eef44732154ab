"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/fpng_amd.h
declares, and its host-side pieces (format tables, checksums, combine) are right.  No GPU compute."""
import os
import re
import zlib

import numpy as np
import pytest

from cpu_ref import ROOT, oracle


def test_library_exports_every_declared_symbol(built_lib):
    from fpng_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "fpng_amd.h")).read()
    declared = set(re.findall(r"\b(fpng_amd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fpng_amd.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert lib.fpng_amd_abi_version() == 2


def test_format_tables_self_check(built_lib):
    import fpng_amd
    assert fpng_amd.layout_1pass(4) == (490, 12, 61)   # reference fpng.cpp:548-551
    assert fpng_amd.layout_1pass(3) == (503, 12, 62)   # reference fpng.cpp:532-535
    # the oracle derives the same layout independently
    for c in (3, 4):
        lens, codes, prefix, sbit = oracle().table_1pass(c)
        assert sbit == fpng_amd.layout_1pass(c)[0] and lens[256] == 12


def test_host_checksums(built_lib):
    import fpng_amd
    rng = np.random.default_rng(1)
    for n in (0, 1, 3, 17, 5552, 5553, 70000, 1 << 21):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        assert fpng_amd.fpng_crc32(d) == zlib.crc32(d.tobytes())
        assert fpng_amd.fpng_adler32(d) == zlib.adler32(d.tobytes())
        k = n // 3
        assert fpng_amd.fpng_crc32(d[k:], fpng_amd.fpng_crc32(d[:k])) == zlib.crc32(d.tobytes())
        assert fpng_amd.fpng_adler32(d[k:], fpng_amd.fpng_adler32(d[:k])) == zlib.adler32(d.tobytes())
        assert fpng_amd.crc32_combine(zlib.crc32(d[:k].tobytes()), zlib.crc32(d[k:].tobytes()), n - k) == zlib.crc32(d.tobytes())
        assert fpng_amd.adler32_combine(zlib.adler32(d[:k].tobytes()), zlib.adler32(d[k:].tobytes()), n - k) == zlib.adler32(d.tobytes())


def test_max_encoded_size(built_lib):
    import fpng_amd
    for (w, h, c) in [(1, 1, 3), (1, 1, 4), (512, 512, 3), (3840, 2160, 4), (65535, 1, 3), (21845, 1, 3), (21846, 1, 3)]:
        assert fpng_amd.max_encoded_size(w, h, c) == oracle().max_size(w, h, c)
        n = (w * c + 1) * h
        assert fpng_amd.max_encoded_size(w, h, c) == 58 + 6 + n + 5 * ((n + 65534) // 65535) + 16


def test_no_silent_cpu_fallback(built_lib):
    """Without a GPU the encode path must fail loudly, never produce bytes some other way."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import fpng_amd
    img = np.zeros((4, 4, 3), dtype=np.uint8)
    with pytest.raises(Exception):
        fpng_amd.fpng_encode_image_to_memory(img, 4, 4, 3)
    assert fpng_amd.fpng_cpu_supports_sse41() is False
