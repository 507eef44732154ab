"""Screenshot-like content (tests/ui_images.py: glyph rows, flat panels with anti-aliased edges, ordered-dither gradients, alpha
mattes with exact runs across the 256-pixel super-windows) against the golden vectors the unmodified reference produced
(tests/golden/ui.json, oracle/make_golden_ui.py): the C restatement on the CPU; on the GPU every encode path (submission, host
stream, 8 row bands, node image) and the GPU decoder's round trip (host files, device-resident files)."""
import hashlib
import json
import os

import numpy as np
import pytest

import ui_images
from cpu_ref import ROOT, oracle


def gold():
    with open(os.path.join(ROOT, "tests", "golden", "ui.json")) as f:
        return json.load(f)


def test_generators_and_the_c_restatement_match_the_golden_vectors():
    g = gold()
    imgs = ui_images.all_images()
    assert set(imgs) == set(g)
    for name, (img, w, h, c) in imgs.items():
        assert hashlib.sha256(img.tobytes()).hexdigest() == g[name]["pixels_sha256"], name
        if w > 1920:
            continue  # (the 4K ones go through the GPU below; the oracle is a scalar encoder)
        for flags in (0, 1):
            png = oracle().encode(img, w, h, c, flags)
            assert len(png) == g[name]["flags"][str(flags)]["size"] and hashlib.sha256(png).hexdigest() == g[name]["flags"][str(flags)]["sha256"], (name, flags)


@pytest.mark.gpu
def test_every_encode_path_and_the_decoder_on_ui_content(built_lib):
    import torch
    import fpng_amd
    from fpng_amd import sharded
    g = gold()
    imgs = ui_images.all_images()
    enc = fpng_amd.Encoder(device=0)
    node = fpng_amd.Node([0, 0, 0])
    try:
        names = sorted(imgs)
        ts = [torch.from_numpy(imgs[n][0]).cuda() for n in names]
        for flags in (0, 1):
            pngs, modes = enc.encode_tensors(ts, flags)  # one submission
            for n, png in zip(names, pngs):
                e = g[n]["flags"][str(flags)]
                assert len(png) == e["size"] and hashlib.sha256(png).hexdigest() == e["sha256"], (n, flags, "submission")
            for n, t, png in zip(names, ts, pngs):
                img, w, h, c = imgs[n]
                if flags == 0:
                    assert enc.encode_host(img, w, h, c, 0) == png, (n, "host stream")
                cuts = [b[0] for b in sharded.split_rows(h, 8)] + [h]
                assert sharded.encode_image_bands_local(sharded.GpuBandBackend(enc), t, cuts, flags) == png, (n, flags, "8 bands")
                assert node.encode_host_image(img, w, h, c, flags) == png, (n, flags, "node image")
            # the way back: host files in one batch, then the same files device-resident
            for got in (enc.decode_batch(pngs, 4), enc.decode_device([torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda() for p in pngs], 4,
                                                                     [(imgs[n][1], imgs[n][2]) for n in names])):
                for n, t, (st, px, cf) in zip(names, ts, got):
                    assert st == 0 and cf == t.shape[2], (n, flags, st)
                    want = t if t.shape[2] == 4 else torch.cat([t, torch.full_like(t[:, :, :1], 255)], dim=2)
                    assert torch.equal(px, want), (n, flags, "decode")
    finally:
        node.close()
        enc.close()
