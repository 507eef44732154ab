#!/usr/bin/env python
"""fpng::fpng_decode_memory() through libfpng.so, host PNG -> host pixels into a reused std::vector, timed in C++: the GPU tier
(images of 256K pixels and more; upload, decode, download included) next to the CPU tier (FPNG_AMD_DECODE_CPU=1, a fresh process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:  # child: print seconds per call for every case
    import numpy as np, torch, fpng_amd, dropin, real_image
    enc = fpng_amd.Encoder(device=0)
    imgs = real_image.variants(real_image.rgb_pixels(dropin.decode))
    cases = [("photo 687x1012 RGB", imgs["rgb"]), ("photo 2748x4048 RGB", imgs["rgb_t4"]), ("1080p RGB grad", fpng_amd.synth_image("grad", 1920, 1080, 3)),
             ("4K RGBA grad", fpng_amd.synth_image("grad", 3840, 2160, 4)), ("8K RGBA grad", fpng_amd.synth_image("grad", 7680, 4320, 4))]
    for name, img in cases:
        (png,), _ = enc.encode_tensors([torch.from_numpy(img).cuda()], 0)
        t = dropin.time_decode(png, 4, reps=4)
        tr = 0.0
        if sys.argv[1] == "gpu":  # the reference's own decoder (oracle/_ref: the unmodified fpng.cpp, SSE4.1 build) on one core, into a reused buffer
            import time, ctypes as C
            from cpu_ref import ref
            R = ref(); b = np.frombuffer(png, dtype=np.uint8); out = np.zeros(img.shape[0] * img.shape[1] * 4, dtype=np.uint8)
            w, h, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
            tr = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); st = R.L.ref_decode(b.ctypes.data, b.size, out.ctypes.data, out.size, C.byref(w), C.byref(h), C.byref(c), 4); tr = min(tr, time.perf_counter() - t0)
            assert st == 0
        print(f"{name}|{img.shape[0] * img.shape[1]}|{len(png)}|{t}|{tr}", flush=True)
    sys.exit(0)
res = {}
for tier, env in (("gpu", {}), ("cpu", {"FPNG_AMD_DECODE_CPU": "1"})):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), tier], env=dict(os.environ, **env), capture_output=True, text=True).stdout
    for ln in out.splitlines():
        f = ln.split("|")
        if len(f) == 5:
            res.setdefault(f[0], {})[tier] = (int(f[1]), int(f[2]), float(f[3]), float(f[4]))
for name, r in res.items():
    px, nb, tg, tr = r["gpu"]
    tc = r["cpu"][2]
    print(f"{name} ({nb / 1e6:.1f} MB of PNG): GPU tier {tg * 1e3:8.2f} ms = {px / tg / 1e6:9.1f} MP/s | CPU tier {tc * 1e3:8.2f} ms = {px / tc / 1e6:7.1f} MP/s | "
          f"the reference's decoder {tr * 1e3:8.2f} ms = {px / tr / 1e6:7.1f} MP/s | GPU tier x {tr / tg:.1f} the reference")
