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
    assert lib.fpng_amd_abi_version() == 5


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


def test_host_checksums_every_length_and_alignment(built_lib):
    """The host CRC-32 switches between a carry-less-multiply body (64 bytes and more, multiples of 16), slicing-by-16 and single
    bytes, the Adler-32 between 16-byte vector blocks and single bytes, reducing every 5552 bytes: every length up to 400, lengths
    around the chunk borders, odd alignments, non-trivial previous values, all-0xFF input (the largest sums) -- against zlib."""
    import fpng_amd
    rng = np.random.default_rng(2)
    big = rng.integers(0, 256, 40000, dtype=np.uint8)
    for n in list(range(0, 400)) + [5551, 5552, 5553, 5567, 5568, 5569, 11103, 11104, 11105, 16383, 16384, 16385, 39999]:
        for off in (0, 1, 7):
            d = big[off:off + n]
            assert fpng_amd.fpng_crc32(d) == zlib.crc32(d.tobytes()), (n, off)
            assert fpng_amd.fpng_adler32(d) == zlib.adler32(d.tobytes()), (n, off)
            assert fpng_amd.fpng_crc32(d, 0xDEADBEEF) == zlib.crc32(d.tobytes(), 0xDEADBEEF), (n, off)
            assert fpng_amd.fpng_adler32(d, 0xFFF0FFEF) == zlib.adler32(d.tobytes(), 0xFFF0FFEF), (n, off)  # (both halves 65520 - ...: near the modulus)
    ff = np.full(100000, 255, dtype=np.uint8)
    for n in (5552, 5553, 65536, 100000):
        assert fpng_amd.fpng_adler32(ff[:n]) == zlib.adler32(ff[:n].tobytes())
        assert fpng_amd.fpng_adler32(ff[:n], zlib.adler32(ff[:777].tobytes())) == zlib.adler32(ff[:n].tobytes(), zlib.adler32(ff[:777].tobytes()))
        assert fpng_amd.fpng_crc32(ff[:n]) == zlib.crc32(ff[:n].tobytes())


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


def test_decode_entry_points_reject_bad_arguments_before_touching_the_gpu(built_lib):
    """fpng_amd_decode_batch / fpng_amd_decode_host: NULL encoder, NULL result, wrong channel count -> FPNG_AMD_ERR_INVALID_ARG
    (no HIP call is made on these ways out, so this runs without a GPU)."""
    import ctypes as C
    from fpng_amd import _lib
    lib = _lib.load()
    res = _lib.DecodeResult()
    png = (C.c_uint8 * 64)()
    cb = _lib.RESERVE_FN(lambda user, n: None)
    assert lib.fpng_amd_decode_host(None, png, 64, 4, cb, None, C.byref(res)) < 0
    assert lib.fpng_amd_decode_batch(None, None, 0, 4, None) < 0
    assert "null" in lib.fpng_amd_last_error().decode().lower() or lib.fpng_amd_last_error()


def test_decode_descriptor_reads_its_result_records_without_a_loop(built_lib):
    """fpng_amd.api.DecodeBatch (Encoder.make_decode_batch): statuses() is a view of the fpng_amd_decode_result records the C call
    fills -- field order and size as include/fpng_amd.h declares them -- and results() hands out views of the output tensors."""
    import ctypes as C
    import numpy as np
    import torch
    from fpng_amd import _lib
    from fpng_amd.api import DecodeBatch
    assert C.sizeof(_lib.DecodeResult) == 16 and _lib.DecodeResult.status.offset == 12
    n = 5
    res = (_lib.DecodeResult * n)()
    outs = [torch.arange(6 * 4 * 3, dtype=torch.uint8) for _ in range(n)]
    for i in range(n):
        res[i].w, res[i].h, res[i].channels_in_file, res[i].status = 6, 4, 3 + (i & 1), (0, 3, 64, 0, 1)[i]
    db = DecodeBatch([None] * n, outs, (_lib.PngIn * n)(), res, 3)
    assert list(db.statuses()) == [0, 3, 64, 0, 1]
    res[1].status = 0  # (a view, not a copy)
    assert list(db.statuses()) == [0, 0, 64, 0, 1]
    got = db.results()
    assert [st for st, _, _ in got] == [0, 0, 64, 0, 1] and [cf for _, _, cf in got] == [3, 4, 3, 4, 3]
    assert got[2][1] is None and got[4][1] is None and tuple(got[0][1].shape) == (4, 6, 3)
    assert np.array_equal(got[3][1].numpy().reshape(-1), np.arange(72, dtype=np.uint8))


def _raw_crc(data):
    """CRC-32 with init 0 and no final xor (what the kernels' partials are made of): crc32 is affine in the message."""
    return zlib.crc32(data) ^ zlib.crc32(bytes(len(data)))


def test_band_plan_matches_the_python_mirror(built_lib):
    """fpng_amd_plan_bands / fpng_amd_band_window (C, what C++ callers and the streamed host path use) against
    fpng_amd/sharded.py's plan_bands / window_extent on random band records, incl. empty bands and stored outcomes."""
    import fpng_amd
    from fpng_amd import _lib, sharded
    rng = np.random.default_rng(5)
    n_stored = 0
    for trial in range(300):
        c = int(rng.integers(3, 5))
        w, h = int(rng.integers(1, 5000)), int(rng.integers(1, 3000))
        nb = int(rng.integers(1, 9))
        cuts = sorted(int(v) for v in rng.integers(0, h + 1, nb - 1))
        cuts = [0] + cuts + [h]
        one_pass = bool(rng.integers(0, 2))
        ftb = fpng_amd.layout_1pass(c)[0] if one_pass else int(rng.integers(300, 900))
        scale = float(rng.choice([0.3, 0.9, 1.0, 1.01, 1.2]))
        py, cs = [], []
        for y0, y1 in zip(cuts[:-1], cuts[1:]):
            nbytes = (w * c + 1) * (y1 - y0)
            bits = int(nbytes * 8 * scale * rng.random()) if nbytes else 0
            s1, s2, lu = (int(rng.integers(0, 65521)), int(rng.integers(0, 65521)), int(rng.integers(1, 60))) if nbytes else (0, 0, 0)
            py.append(sharded.BandStats(bits, s1, s2, nbytes, lu, ftb, 12))
            cs.append(_lib.BandStats(bits, s1, s2, nbytes, lu, ftb, 12, 0))
        want = sharded.plan_bands(py, w, h, c, ftb, 12, one_pass)
        starts, plan = fpng_amd.plan_bands(cs, w, h, c, 0 if one_pass else 1)
        assert starts == want.start_bits and plan.end_bit == want.end_bit and plan.adler == want.adler
        assert bool(plan.stored) == want.stored and plan.zlib_size == want.zlib_size
        n_stored += want.stored
        non_empty = [k for k in range(nb) if py[k].nbytes]
        for k in non_empty:
            first, last = k == non_empty[0], cuts[k + 1] == h
            off, nbytes, head = fpng_amd.band_window(first, last, starts[k], py[k].token_bits, 12)
            assert (off, nbytes) == sharded.window_extent(first, last, starts[k], py[k].token_bits, 12)
            assert head == (0 if first else (min(16, nbytes) if (58 * 8 + starts[k]) % 128 else 0))
    assert 0 < n_stored < 300


def test_container_from_band_crcs(built_lib):
    """fpng_amd_idat_crc_from_bands / fpng_amd_png_head / fpng_amd_png_tail rebuild a reference file's container from raw CRCs
    of arbitrary windows of its zlib stream (what the streamed host path and the sharded path do with the GPU's values)."""
    import fpng_amd
    rng = np.random.default_rng(6)
    for (kind, w, h, c) in [("grad", 97, 41, 3), ("blocks", 300, 77, 4), ("grad", 640, 360, 4)]:
        img = fpng_amd.synth_image(kind, w, h, c)
        png = oracle().encode(img, w, h, c, 0)
        zlib_size = len(png) - 58 - 16
        adler = int.from_bytes(png[58 + zlib_size - 4:58 + zlib_size], "big")
        assert fpng_amd.png_head(w, h, c, zlib_size) == png[:58]
        data_end = 58 + zlib_size - 4
        for trial in range(10):
            nb = int(rng.integers(1, 6))
            # windows end on 16-byte pieces of the FILE (the last one may end behind the data, zero padded)
            ends = sorted(set(int(v) & ~15 for v in rng.integers(80, data_end, nb - 1))) + [(data_end + 15) & ~15]
            raw, begin = [], 58
            for e in ends:
                seg = png[begin:min(e, data_end)]
                # raw CRC of bytes [58, e) with everything outside [begin, e) zero
                raw.append(_raw_crc(bytes(begin - 58) + seg + bytes(e - begin - len(seg))))
                begin = e
            crc = fpng_amd.idat_crc_from_bands(raw, ends, zlib_size, adler)
            assert fpng_amd.png_tail(adler, crc) == png[58 + zlib_size - 4:]


def test_bench_reads_its_own_pmc_summary_and_prints_no_rate_above_the_peak():
    """Round 4's driver line carried chain_traffic_rate_GBs = 13 244 GB/s: bench.py's glob for the encode command's PMC summary
    (`*_8k_pmc_traffic.txt`) also matched the decoder's (`r04_decode_8k_pmc_traffic.txt`).  The finder matches whole basenames
    now, and no HBM rate above the peak leaves the script."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    enc_files = [os.path.basename(p) for p in bench._pmc_summaries("8k")]
    dec_files = [os.path.basename(p) for p in bench._pmc_summaries("decode_8k")]
    assert enc_files and dec_files and not set(enc_files) & set(dec_files)
    assert all("decode" not in n and "2pass" not in n and "noise" not in n for n in enc_files)
    assert all(n.split("_", 1)[1].startswith("decode_8k_") for n in dec_files)
    chain = bench.committed_chain_traffic("8k")
    alg = 8 * 7680 * 4320 * 4 + 464_000_000
    assert alg < chain < 2.5 * alg  # the encode chain's, not the decoder's 6.3 GB
    line = {"roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achieved": 4300.0, "chain_traffic_rate_GBs": 13244.0},
            "decode": {"roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achieved": 1600.0}}}
    errs = bench.check_rates(line)
    assert len(errs) == 1 and "chain_traffic_rate_GBs" in errs[0] and line["roofline"]["chain_traffic_rate_GBs"] is None
    assert line["roofline"]["achieved"] == 4300.0 and bench.scope_name(1) == "one GPU"


def test_loading_the_library_asks_for_eight_hardware_queues_unless_told_otherwise(built_lib):
    """csrc/api.cpp runtime_defaults(): GPU_MAX_HW_QUEUES=8 is put into the process environment when the library is loaded
    (the HIP runtime reads it at its first call), never over a value that is there, never with FPNG_AMD_KEEP_HW_QUEUES=1."""
    import subprocess
    import sys
    prog = ("import ctypes, fpng_amd; libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "print('Q=%s' % (libc.getenv(b'GPU_MAX_HW_QUEUES') or b'-').decode())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra):
        env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "FPNG_AMD_KEEP_HW_QUEUES")}
        env.update(extra, PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return [ln for ln in out.stdout.splitlines() if ln.startswith("Q=")][-1]

    assert run({}) == "Q=8"
    assert run({"GPU_MAX_HW_QUEUES": "4"}) == "Q=4"
    assert run({"FPNG_AMD_KEEP_HW_QUEUES": "1"}) == "Q=-"
