"""Natural-image parity (-m gpu): the one real picture the reference ships (example.png, 687x1012 RGB) and the variants
its harness derives from a source image (32 bpp, -a green->alpha; fpng_test.cpp:1116-1190), plus 4x4 tilings that cross many
256-pixel super-windows -- through every way into the HIP path: device-resident submission, the host-batch front door, row
bands, and the `namespace fpng` drop-in.  Expected bytes: the UNMODIFIED reference (tests/golden/real.json)."""
import hashlib

import numpy as np
import pytest

import dropin
import real_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc(built_lib):
    import torch
    import fpng_amd
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    e = fpng_amd.Encoder(device=0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def imgs():
    return real_image.variants(real_image.rgb_pixels(dropin.decode))


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def test_fixture_is_what_the_hip_two_pass_encoder_writes(enc, imgs):
    import torch
    (png,), _ = enc.encode_tensors([torch.from_numpy(imgs["rgb"]).cuda()], 1)
    assert png == real_image.fixture_bytes()


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_all_variants_one_submission(enc, imgs, flags):
    import torch
    g = real_image.gold()["variants"]
    names = list(imgs)
    pngs, modes = enc.encode_tensors([torch.from_numpy(imgs[k]).cuda() for k in names], flags)
    for k, p, m in zip(names, pngs, modes):
        e = g[k]["flags"][str(flags)]
        assert len(p) == e["size"], f"{k} flags={flags}: {len(p)} bytes, reference {e['size']}"
        assert _sha(p) == e["sha256"], f"{k} flags={flags} differs from the reference"
        assert m == (1 if flags == 2 else 0)


@pytest.mark.parametrize("flags", [0, 1])
def test_host_batch_front_door(enc, imgs, flags):
    import fpng_amd
    g = real_image.gold()["variants"]
    names = ["rgb", "rgba_ga", "rgb_t4", "rgba_ga_t4", "rgba"]
    outs = [np.empty(fpng_amd.max_encoded_size(imgs[k].shape[1], imgs[k].shape[0], imgs[k].shape[2]), dtype=np.uint8) for k in names]
    sizes = enc.encode_host_batch([imgs[k] for k in names], flags, outs=outs)
    for k, o, n in zip(names, outs, sizes):
        assert _sha(o[:n].tobytes()) == g[k]["flags"][str(flags)]["sha256"], f"{k} flags={flags}"


@pytest.mark.parametrize("name", ["rgb", "rgba_ga", "rgb_t4", "rgba_ga_t4"])
@pytest.mark.parametrize("flags", [0, 1])
def test_eight_row_bands(enc, imgs, name, flags):
    import torch
    from fpng_amd import sharded
    img = imgs[name]
    h = img.shape[0]
    cuts = [b[0] for b in sharded.split_rows(h, 8)] + [h]
    png = sharded.encode_image_bands_local(sharded.GpuBandBackend(enc), torch.from_numpy(img).cuda(), cuts, flags)
    assert _sha(png) == real_image.gold()["variants"][name]["flags"][str(flags)]["sha256"]


def test_uneven_bands_cut_inside_the_picture(enc, imgs):
    import torch
    from fpng_amd import sharded
    img = imgs["rgba_ga"]
    h = img.shape[0]
    png = sharded.encode_image_bands_local(sharded.GpuBandBackend(enc), torch.from_numpy(img).cuda(), [0, 1, 2, 257, 258, 700, h - 1, h], 0)
    assert _sha(png) == real_image.gold()["variants"]["rgba_ga"]["flags"]["0"]["sha256"]


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_cpp_dropin(imgs, flags):
    g = real_image.gold()["variants"]
    for k in ("rgb", "rgba_ga"):
        h, w, c = imgs[k].shape
        png = dropin.encode(imgs[k], w, h, c, flags)
        assert _sha(png) == g[k]["flags"][str(flags)]["sha256"], f"{k} flags={flags}"
        st, px, dw, dh, dc = dropin.decode(png, c)
        assert st == 0 and (dw, dh, dc) == (w, h, c) and np.array_equal(px, imgs[k].reshape(-1))
