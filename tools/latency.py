"""Unpipelined single-submission latency: host wall clock from submit() to finish() returning, device-resident frames,
one submission in flight at a time (what a GPU-resident producer that needs each PNG before going on would see)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpng_amd

enc = fpng_amd.Encoder(device=0, stream="own")
for (w, h, c) in [(512, 512, 3), (1920, 1080, 3), (3840, 2160, 4), (7680, 4320, 4)]:
    for flags in (0, 1):
        img = torch.from_numpy(fpng_amd.synth_image("grad", w, h, c)).cuda()
        out = torch.empty(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device="cuda")
        batch = enc.make_batch([img], [out])
        for _ in range(30):
            enc.submit(batch, None, flags); enc.finish(1)
        ts = []
        for _ in range(200):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            enc.submit(batch, None, flags)
            enc.finish(1)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"{w}x{h}x{c} flags={flags}: median {ts[100] * 1e6:7.1f} us, best {ts[0] * 1e6:7.1f} us")
