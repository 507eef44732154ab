#!/usr/bin/env python
"""Host-side cost of one fpng_amd_encode_batch_async() call (descriptor array prebuilt) for a few batch shapes."""
import sys, time, torch
sys.path.insert(0, ".")
import fpng_amd
for (w, h, c, B) in [(7680, 4320, 4, 8), (1920, 1080, 3, 256), (512, 512, 4, 1024)]:
    imgs = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, c, seed=i)).cuda() for i in range(B)]
    cap = fpng_amd.max_encoded_size(w, h, c) + 64
    outs = [[torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(B)] for _ in range(2)]
    enc = fpng_amd.Encoder(device=0, stream="own")
    bs = [enc.make_batch(imgs, o) for o in outs]
    for i in range(4):
        enc.submit(bs[i & 1]); enc.finish(B)
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); enc.submit(bs[i & 1]); ts.append(time.perf_counter() - t0)
        enc.finish(B)
    t0 = time.perf_counter()
    for i in range(20):
        enc.submit(bs[i & 1])
    t1 = time.perf_counter(); enc.finish(B); t2 = time.perf_counter()
    print(f"{w}x{h}x{c} B={B}: submit on idle encoder {1e3*min(ts):.3f} ms; 20 back-to-back submits took {1e3*(t1-t0):.2f} ms, +finish {1e3*(t2-t1):.2f} ms")
    enc.close()
