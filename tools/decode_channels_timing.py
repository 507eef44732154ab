import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, fpng_amd
enc = fpng_amd.Encoder(device=0)
for (w, h, c, n) in ((1920, 1080, 3, 64), (3840, 2160, 4, 16), (1921, 1081, 3, 64), (1923, 1080, 4, 64)):
    ts = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, c, seed=12345 + i)).cuda() for i in range(n)]
    pngs, _ = enc.encode_tensors(ts, 0)
    dev = [torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda() for p in pngs]
    for desired in (3, 4):
        outs = [torch.empty(w * h * desired, dtype=torch.uint8, device="cuda") for _ in range(n)]
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); got = enc.decode_device(dev, desired, [(w, h)] * n, outs); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        enc.set_profiling(True); enc.decode_device(dev, desired, [(w, h)] * n, outs); ph = enc.last_decode_phase_ms(); enc.set_profiling(False)
        print(f"{n} x {w}x{h}x{c} -> {desired} channels: {best*1e3:.3f} ms per step ({sum(len(p) for p in pngs)/1e6:.0f} MB of PNG -> {n*w*h*desired/1e6:.0f} MB); phases " + " ".join(f"{k} {v:.3f}" for k, v in ph.items()), flush=True)
