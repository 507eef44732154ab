"""Host-buffer pipelines (-m gpu): fpng_amd_encode_host_to() streamed in row bands (upload | encode + place | download
overlapped), the fpng:: drop-in on top of it, and the whole-node form -- all against the reference's bytes."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import dropin
import real_image
from cpu_ref import ROOT, oracle

pytestmark = pytest.mark.gpu


def _kat(kind, w, h, c, flags):
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        for e in json.load(f):
            if (e["w"], e["h"], e["c"], e["kind"]) == (w, h, c, kind):
                return e["flags"][str(flags)]
    raise KeyError((kind, w, h, c))


@pytest.fixture(scope="module")
def enc(built_lib):
    import torch
    import fpng_amd
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    e = fpng_amd.Encoder(device=0, stream="own")
    yield e
    e.close()


@pytest.mark.parametrize("dims", [(7680, 4320, 4), (3840, 2160, 4)])
def test_streamed_frames_vs_reference(enc, dims):
    """8K RGBA (8 bands) and 4K RGBA (3 bands by the size rule): golden sha256 of the unmodified reference.  Pageable and
    page-locked buffers alike are streamed from the first call on."""
    import fpng_amd
    w, h, c = dims
    out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
    for kind in ("grad", "blocks"):
        img = fpng_amd.synth_image(kind, w, h, c)
        exp = _kat(kind, w, h, c, 0)
        for call in range(3):
            n = enc.encode_host_into(img, w, h, c, out, 0)
            assert n == exp["size"] and hashlib.sha256(out[:n].tobytes()).hexdigest() == exp["sha256"], (kind, dims, call)
            assert enc.last_host_bands() > 1, (kind, call, enc.last_host_bands())
        png, asked = enc.encode_host_growing(img, w, h, c, 0)      # an output that grows: an estimate first, the exact size last
        assert asked[-1] == exp["size"] and max(asked) <= out.size and hashlib.sha256(png).hexdigest() == exp["sha256"]
        assert enc.encode_host(img, w, h, c, 0) == png
        img2 = img.copy()                                          # a page-locked frame
        fpng_amd.pin_host_memory(img2)
        try:
            n = enc.encode_host_into(img2, w, h, c, out, 0)
            assert enc.last_host_bands() > 1 and hashlib.sha256(out[:n].tobytes()).hexdigest() == exp["sha256"]
        finally:
            fpng_amd.unpin_host_memory(img2)


def test_streamed_frame_into_a_buffer_of_exactly_the_files_size(enc):
    """fpng_amd_encode_host() asks only for out_cap >= the PNG's size: a buffer that is smaller than the streamed path's own
    first estimate of the file (band 0's share extrapolated, plus a margin) but holds the file must do."""
    import fpng_amd
    w, h, c = 7680, 4320, 4
    img = fpng_amd.synth_image("grad", w, h, c)
    exp = _kat("grad", w, h, c, 0)
    for cap in (exp["size"], exp["size"] + 1000):
        out = np.empty(cap, dtype=np.uint8)
        n = enc.encode_host_into(img, w, h, c, out, 0)
        assert enc.last_host_bands() > 1
        assert n == exp["size"] and hashlib.sha256(out[:n].tobytes()).hexdigest() == exp["sha256"]
    with pytest.raises(fpng_amd.FpngAmdError):
        enc.encode_host_into(img, w, h, c, np.empty(exp["size"] - 1, dtype=np.uint8), 0)


def test_streamed_incompressible_frame_ends_up_stored(enc):
    import fpng_amd
    w, h, c = 3840, 2160, 4
    img = fpng_amd.synth_image("noise", w, h, c)
    exp = _kat("noise", w, h, c, 0)
    out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
    fpng_amd.pin_host_memory(img)
    try:
        for _ in range(2):
            n = enc.encode_host_into(img, w, h, c, out, 0)
            assert exp["btype"] == 0 and n == exp["size"] and hashlib.sha256(out[:n].tobytes()).hexdigest() == exp["sha256"]
    finally:
        fpng_amd.unpin_host_memory(img)


def test_half_compressible_frame_crosses_the_budget_mid_stream(enc):
    """Rows that compress first, noise afterwards: the outcome is only known several bands into the stream."""
    import fpng_amd
    w, h, c = 4096, 2048, 4
    out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
    for frac in (3, 20):  # barely compressible / clearly stored
        img = fpng_amd.synth_image("grad", w, h, c).copy()
        img[h // frac:] = fpng_amd.synth_image("noise", w, h - h // frac, c)
        exp = oracle().encode(img, w, h, c, 0)
        for _ in range(3):
            n = enc.encode_host_into(img, w, h, c, out, 0)
            assert out[:n].tobytes() == exp


def test_band_counts_and_the_natural_image(built_lib):
    """FPNG_AMD_HOST_BANDS = 1, 2, 5, 16 in fresh processes (the knob is read once): every cut of the tiled photograph and
    of an RGB frame whose rows do not divide evenly gives the reference's file."""
    code = r'''
import hashlib, sys
sys.path.insert(0, "tests")
import numpy as np
import fpng_amd, dropin, real_image
imgs = real_image.variants(real_image.rgb_pixels(dropin.decode))
g = real_image.gold()["variants"]
enc = fpng_amd.Encoder(device=0, stream="own")
for k in ("rgb_t4", "rgba_ga_t4", "rgb"):
    h, w, c = imgs[k].shape
    for fl in (0, 1, 2):
        png, asked = enc.encode_host_growing(imgs[k], w, h, c, fl)
        assert hashlib.sha256(png).hexdigest() == g[k]["flags"][str(fl)]["sha256"], (k, fl)
print("ok")
'''
    for nb in (1, 2, 5, 16):
        env = dict(os.environ, FPNG_AMD_HOST_BANDS=str(nb))
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "ok" in out.stdout, (nb, out.stderr[-800:])


def test_cpp_dropin_first_and_later_calls(built_lib):
    import fpng_amd
    w, h, c = 7680, 4320, 4
    img = fpng_amd.synth_image("grad", w, h, c)
    exp = _kat("grad", w, h, c, 0)
    for _ in range(3):  # (the shim's vector is new every call; the timing shim keeps one)
        png = dropin.encode(img, w, h, c, 0)
        assert len(png) == exp["size"] and hashlib.sha256(png).hexdigest() == exp["sha256"]
    t, n = dropin.time_encode(img, w, h, c, 0, reps=4, reuse=True)
    assert n == exp["size"]
    exp1 = _kat("grad", w, h, c, 1)
    assert hashlib.sha256(dropin.encode(img, w, h, c, 1)).hexdigest() == exp1["sha256"]


def test_node_host_batch_two_pipelines_on_one_gpu(built_lib, tmp_path):
    """fpng_amd_node_*: device 0 listed twice = two encoders with their own rings and threads; frames dealt round-robin,
    to memory and to files."""
    import fpng_amd
    node = fpng_amd.Node([0, 0])
    assert node.size() == 2
    frames = [fpng_amd.synth_image("grad", 1920, 1080, 3, seed=12345 + i) for i in range(7)]
    with open(os.path.join(ROOT, "tests", "golden", "batches.json")) as f:
        g = json.load(f)["c3"]["flags"]["0"]
    outs = [np.empty(fpng_amd.max_encoded_size(1920, 1080, 3), dtype=np.uint8) for _ in frames]
    paths = [str(tmp_path / f"n{i}.png") for i in range(len(frames))]
    sizes = node.encode_host_batch(frames, 0, outs=outs, paths=paths, writer_threads=2)
    for i, (o, n) in enumerate(zip(outs, sizes)):
        assert n == g["sizes"][i] and hashlib.sha256(o[:n].tobytes()).hexdigest() == g["sha256"][i]
        assert open(paths[i], "rb").read() == o[:n].tobytes()
    node.close()


def test_copy_streams_end_up_on_two_engines(built_lib):
    """The host paths' upload and download streams must not share an SDMA engine (pipeline.cpp: ensure_copy_streams) -- with one
    engine every download queues behind the uploads and a streamed 8K call costs 3.5 instead of 2.8 ms.  Fresh process: the pair
    is made once per encoder; FPNG_AMD_TRACE prints what the library measured when it made it."""
    code = r'''
import numpy as np, fpng_amd
w, h, c = 7680, 4320, 4
img = fpng_amd.synth_image("grad", w, h, c)
out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
enc = fpng_amd.Encoder(device=0, stream="own")
for _ in range(2):
    n = enc.encode_host_into(img, w, h, c, out, 0)
print("bands", enc.last_host_bands(), "bytes", n)
'''
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, FPNG_AMD_TRACE="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-800:]
    made = [l for l in out.stderr.splitlines() if "copy streams, attempt" in l]
    assert made and made[-1].endswith("two engines"), made
    assert int(out.stdout.split()[1]) >= 8


def test_cpp_dropin_called_from_many_threads_at_once(built_lib):
    """SURVEY 8b threading: the reference's encoder is re-entrant and callers encode from many threads.  Six threads, six
    different frames (streamed 8K / 4K frames, small ones, both channel counts, all three flag values), three calls each,
    all through fpng::fpng_encode_image_to_memory at the same time; every file equals the CPU checker's."""
    import fpng_amd
    specs = [("grad", 7680, 4320, 4, 0), ("blocks", 3840, 2160, 4, 0), ("grad", 1920, 1080, 3, 1), ("noise", 640, 480, 4, 0),
             ("grad", 3840, 2160, 3, 0), ("grad", 333, 77, 3, 2)]
    imgs = [fpng_amd.synth_image(k, w, h, c, seed=100 + i) for i, (k, w, h, c, _) in enumerate(specs)]
    pngs, agree = dropin.encode_threads(imgs, [s[4] for s in specs], reps=3)
    assert agree
    for (k, w, h, c, fl), im, png in zip(specs, imgs, pngs):
        assert png == oracle().encode(im, w, h, c, fl), (k, w, h, c, fl)


def test_encoders_one_after_the_other_and_the_buffer_list(built_lib):
    """Device buffers an encoder gives up are kept by the library and serve the next encoder (encoder.h: DeviceBuf);
    fpng_amd_release_cached_memory() empties that list.  Either way the files are the reference's."""
    import fpng_amd
    w, h, c = 3840, 2160, 4
    img = fpng_amd.synth_image("grad", w, h, c)
    exp = _kat("grad", w, h, c, 0)
    out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
    for k in range(4):
        e = fpng_amd.Encoder(device=0, stream="own")
        n = e.encode_host_into(img, w, h, c, out, 0)
        assert n == exp["size"] and hashlib.sha256(out[:n].tobytes()).hexdigest() == exp["sha256"], k
        e.close()
        if k == 1:
            fpng_amd.release_cached_memory()


def test_one_image_over_the_nodes_devices():
    """fpng_amd_node_encode_host_image: ONE host image cut into row bands over the node's devices, every device moving its own band
    up and its own window of the file down.  With device 0 listed 2 / 3 / 8 times (a one-GPU box): the
    config-4 image (16384 x 16384 RGBA) for both flags against the reference's golden sha256, 4K frames, the photograph (a band
    seam in every row-count class), an incompressible frame (the stored outcome goes back to the whole-image path), a tiny one."""
    import fpng_amd
    import real_image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dropin
    with open(os.path.join(ROOT, "tests", "golden", "batches.json")) as f:
        c4 = json.load(f)["c4"]
    nodes = {n: fpng_amd.Node([0] * n) for n in (2, 3, 8)}
    try:
        w, h, c = c4["w"], c4["h"], c4["c"]
        img = fpng_amd.synth_image(c4["kind"], w, h, c, seed=c4["seed0"])
        for flags in (0, 1):
            exp = c4["flags"][str(flags)]
            for n in (8, 2) if flags == 0 else (8,):
                png = nodes[n].encode_host_image(img, w, h, c, flags)
                assert len(png) == exp["sizes"][0] and hashlib.sha256(png).hexdigest() == exp["sha256"][0], (flags, n)
        del img
        for kind in ("grad", "blocks", "noise"):
            img = fpng_amd.synth_image(kind, 3840, 2160, 4)
            for flags in (0, 1):
                exp = _kat(kind, 3840, 2160, 4, flags)
                for n in (3, 8):
                    png = nodes[n].encode_host_image(img, 3840, 2160, 4, flags)
                    assert len(png) == exp["size"] and hashlib.sha256(png).hexdigest() == exp["sha256"], (kind, flags, n)
        rgb = real_image.rgb_pixels(dropin.decode)
        hh, ww, _ = rgb.shape
        for flags in (0, 1):
            want = oracle_png(rgb, ww, hh, 3, flags)
            for n in (2, 3, 8):
                assert nodes[n].encode_host_image(rgb, ww, hh, 3, flags) == want, (flags, n)
        tiny = fpng_amd.synth_image("grad", 37, 5, 3)
        assert nodes[8].encode_host_image(tiny, 37, 5, 3, 0) == oracle_png(tiny, 37, 5, 3, 0)
    finally:
        for nd in nodes.values():
            nd.close()


def oracle_png(img, w, h, c, flags):
    from cpu_ref import have_ref, oracle, ref
    return (ref() if have_ref() else oracle()).encode(img, w, h, c, flags)
