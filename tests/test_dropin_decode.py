"""CPU tests of the decoder half of the `namespace fpng` drop-in (fpng_amd/csrc/fpng_decode.cpp)
against the reference decoder: same pixels, same status codes, including on damaged files."""
import numpy as np
import pytest

import dropin
from cpu_ref import fuzz_image, have_ref, oracle, ref


def test_roundtrip_all_channel_conversions():
    rng = np.random.default_rng(31)
    for _ in range(150):
        img, w, h, c = fuzz_image(rng)
        for fl in (0, 1, 2):
            png = oracle().encode(img, w, h, c, fl)
            assert dropin.get_info(png) == (0, w, h, c)
            for desired in (3, 4):
                st, out, ww, hh, cc = dropin.decode(png, desired)
                assert st == 0 and (ww, hh, cc) == (w, h, c)
                exp = img.reshape(h, w, c)
                if desired == c:
                    exp2 = exp
                elif desired == 3:
                    exp2 = exp[:, :, :3]
                else:
                    exp2 = np.concatenate([exp, np.full((h, w, 1), 255, dtype=np.uint8)], axis=2)
                assert (out == exp2.reshape(-1)).all(), (w, h, c, fl, desired)


@pytest.mark.skipif(not have_ref(), reason="reference build not available")
def test_status_codes_match_reference_on_damaged_files():
    rng = np.random.default_rng(32)
    checked = differing = 0
    for _ in range(120):
        img, w, h, c = fuzz_image(rng)
        png = bytearray(oracle().encode(img, w, h, c, int(rng.integers(0, 3))))
        for _ in range(12):
            bad = bytearray(png)
            kind = rng.integers(0, 4)
            if kind == 0:      # flip a bit anywhere
                i = int(rng.integers(0, len(bad)))
                bad[i] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:    # truncate
                bad = bad[: int(rng.integers(0, len(bad)))]
            elif kind == 2:    # damage the header region
                i = int(rng.integers(0, min(58, len(bad))))
                bad[i] = int(rng.integers(0, 256))
            else:              # damage the zlib stream
                i = int(rng.integers(58, len(bad)))
                bad[i] = int(rng.integers(0, 256))
            bad = bytes(bad)
            if not bad:
                continue
            for desired in (3, 4):
                st_r, out_r, *_ = ref().decode(bad, desired)
                st_m, out_m, *_ = dropin.decode(bad, desired)
                checked += 1
                assert st_r == st_m, (w, h, c, kind, st_r, st_m)
                if st_r == 0:
                    assert (out_r == out_m).all()
    assert checked > 2000


def test_large_image_decodes_with_or_without_a_gpu():
    """Images of 256K pixels and more take the drop-in's GPU tier when the process has a GPU (tests/test_gpu_decode.py holds that
    one against the reference decoder); without one -- this container -- the same call must quietly use the CPU decoder."""
    import fpng_amd
    img = fpng_amd.synth_image("grad", 1024, 768, 3)
    for fl in (0, 1, 2):
        png = oracle().encode(img, 1024, 768, 3, fl)
        for desired in (3, 4):
            st, out, w, h, c = dropin.decode(png, desired)
            assert st == 0 and (w, h, c) == (1024, 768, 3)
            exp = img if desired == 3 else np.concatenate([img, np.full((768, 1024, 1), 255, dtype=np.uint8)], axis=2)
            assert np.array_equal(out, exp.reshape(-1))


def test_not_png_and_bad_args():
    assert dropin.get_info(b"hello world, definitely not a png file at all, padding padding padding padding")[0] == 3  # FAILED_NOT_PNG
    png = oracle().encode(np.zeros((4, 4, 3), dtype=np.uint8), 4, 4, 3, 0)
    st, *_ = dropin.decode(png, 5)
    assert st == 2  # FPNG_DECODE_INVALID_ARG (reference fpng.cpp:3092-3096)


def test_checksum_wrappers():
    import zlib
    d = np.random.default_rng(1).integers(0, 256, 4097, dtype=np.uint8)
    L = dropin.shim()
    assert L.shim_crc32(d.ctypes.data, d.size, 0) == zlib.crc32(d.tobytes())
    assert L.shim_adler32(d.ctypes.data, d.size, 1) == zlib.adler32(d.tobytes())
