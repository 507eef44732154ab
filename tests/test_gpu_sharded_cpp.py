"""One image over several ranks through the C ABI (-m gpu): fpng_amd_encode_image_sharded (fpng_amd/csrc/sharded.cpp).
A one-GPU box cannot run two RCCL ranks, so the multi-rank logic runs over an in-process transport (tests/cpp/sharded_local.cpp:
N threads, N encoders on device 0, device-to-device exchanges) and the built-in RCCL transport runs with one rank."""
import hashlib
import json
import os

import numpy as np
import pytest

import dropin
import real_image
from cpu_ref import ROOT, fuzz_image, oracle

pytestmark = pytest.mark.gpu


def _cuts(h, world):
    base, rem = divmod(h, world)
    cuts = [0]
    for r in range(world):
        cuts.append(cuts[-1] + base + (1 if r < rem else 0))
    return cuts


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("flags", [0, 1])
def test_threads_as_ranks_natural_image(built_lib, world, flags):
    imgs = real_image.variants(real_image.rgb_pixels(dropin.decode))
    g = real_image.gold()["variants"]
    for k in ("rgb", "rgba_ga"):
        h = imgs[k].shape[0]
        png = dropin.encode_sharded_local(imgs[k], _cuts(h, world), flags, root=world - 1 if world > 1 else 0)
        assert hashlib.sha256(png).hexdigest() == g[k]["flags"][str(flags)]["sha256"], (k, world, flags)


def test_ranks_without_rows_root_in_the_middle_and_uneven_bands(built_lib):
    import fpng_amd
    img = fpng_amd.synth_image("grad", 1000, 37, 3)
    exp0, exp1 = oracle().encode(img, 1000, 37, 3, 0), oracle().encode(img, 1000, 37, 3, 1)
    for cuts, root in [([0, 0, 1, 1, 20, 37], 2), ([0, 36, 37, 37], 1), ([0, 5, 5, 5, 37], 3), ([0, 37, 37], 1)]:
        assert dropin.encode_sharded_local(img, cuts, 0, root) == exp0, (cuts, root)
        assert dropin.encode_sharded_local(img, cuts, 1, root) == exp1, (cuts, root)


def test_seams_of_every_kind_fuzz(built_lib):
    """Small edge-case images cut into 2-4 bands: many rows per 16-byte piece, seams on and off piece boundaries,
    stored outcomes (the rows travel to the root)."""
    rng = np.random.default_rng(77)
    stored = 0
    for trial in range(60):
        img, w, h, c = fuzz_image(rng, force_dims=(int(rng.integers(1, 300)), int(rng.integers(2, 30))))
        world = int(rng.integers(2, 5))
        cuts = [0] + sorted(int(v) for v in rng.integers(0, h + 1, world - 1)) + [h]
        flags = int(rng.integers(0, 2))
        exp = oracle().encode(img, w, h, c, flags)
        got = dropin.encode_sharded_local(img, cuts, flags, root=int(rng.integers(0, world)))
        assert got == exp, (trial, w, h, c, cuts, flags)
        stored += exp[60] & 6 == 0
    assert 5 < stored < 55


def test_4k_frame_eight_ranks_vs_reference(built_lib):
    import fpng_amd
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        kat = {(e["kind"], e["w"], e["h"], e["c"]): e for e in json.load(f)}
    for kind in ("grad", "blocks", "noise"):
        e = kat[(kind, 3840, 2160, 4)]
        img = fpng_amd.synth_image(kind, 3840, 2160, 4)
        for flags in (0, 1):
            reports = []
            png = dropin.encode_sharded_local(img, _cuts(2160, 8), flags, root=0, reports=reports)
            assert hashlib.sha256(png).hexdigest() == e["flags"][str(flags)]["sha256"], (kind, flags)
            # where the bytes went (fpng_amd_sharded_last_report): a compressed image's windows travel ONCE, from their ranks straight
            # into the root's file buffer -- no staging copy on the root; the stored outcome (noise) moves the rows instead
            root, others = reports[0], reports[1:]
            if kind == "noise":
                assert root.stored == 1 and root.root_staged_bytes == 3840 * 2160 * 4 and sum(r.sent_bytes for r in others) == 3840 * 2160 * 4 // 8 * 7
            else:
                assert root.stored == 0 and root.root_staged_bytes == 0 and all(r.root_staged_bytes == 0 for r in others)
                assert root.received_in_place + 16 * root.shared_pieces == sum(r.sent_bytes for r in others) and root.sent_bytes == 0
                assert root.own_window_bytes + sum(r.sent_bytes for r in others) >= len(png) - 58 - 20 - 16 * 8  # (the windows are the file's zlib bytes)
                assert root.collectives == (3 if flags else 2)


def test_rccl_transport_one_rank(built_lib):
    """The built-in transport (librccl.so.1 through dlopen; ncclAllGather / ncclAllReduce / group calls on the encoder's
    stream) with world = 1: the whole exchange code runs, the collectives are trivial."""
    import torch
    import fpng_amd
    from fpng_amd import sharded
    enc = fpng_amd.Encoder(device=0)
    sh = sharded.CppRowSharded(enc, 0, 1, sharded.CppRowSharded.rccl_unique_id(), 0)
    imgs = real_image.variants(real_image.rgb_pixels(dropin.decode))
    g = real_image.gold()["variants"]
    for k in ("rgb_t4", "rgba_ga"):
        h, w, c = imgs[k].shape
        rows = torch.from_numpy(imgs[k]).cuda()
        out = torch.empty(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device="cuda")
        for flags in (0, 1):
            png = sh.encode(rows, None, w, h, c, 0, h, flags, 0, out)
            assert hashlib.sha256(png.cpu().numpy().tobytes()).hexdigest() == g[k]["flags"][str(flags)]["sha256"], (k, flags)
    sh.close()
    enc.close()
