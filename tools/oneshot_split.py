"""Eight (or N) device-resident frames in hand, GPU idle: one submission of N, or N/k submissions of k?  (round 5)
Time from the first submit call to the last result, best of 7.
    python tools/oneshot_split.py [w h c N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpng_amd
import torch
w, h, c, N = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (7680, 4320, 4, 8)
imgs = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, c, seed=12345 + i)).cuda() for i in range(N)]
cap = fpng_amd.max_encoded_size(w, h, c) + 64
outs = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(N)]
enc = fpng_amd.Encoder(device=0, stream="own")
for flags in (0, 1):
    for k in [N, N // 2, N // 4, N // 8]:
        if k < 1:
            continue
        batches = [enc.make_batch(imgs[i:i + k], outs[i:i + k]) for i in range(0, N, k)]
        best = 1e9
        for rep in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tickets = []
            for b in batches:
                enc.submit(b, None, flags)
                tickets.append(enc.last_ticket)
            for t in tickets:
                enc.wait(t, k)
            el = time.perf_counter() - t0
            if rep >= 3:
                best = min(best, el)
        print(f"{N} x {w}x{h}x{c} flags={flags}: {N // k} submission(s) of {k}: {best * 1e3:.3f} ms = {N * w * h / best / 1e9:.1f} GP/s", flush=True)
enc.close()
