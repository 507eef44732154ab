"""Where does the first timed region of a short-row bench line lose its time?  (round 5)  Regions like bench.py's -- K submissions back to
back, then finish -- with the host time of every submit call and the completion time of every ticket.
    python tools/region_diag.py w h c B [out_sets] [regions]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpng_amd
import torch
w, h, c, B = (int(v) for v in sys.argv[1:5])
n_sets = int(sys.argv[5]) if len(sys.argv) > 5 else 8
regions = int(sys.argv[6]) if len(sys.argv) > 6 else 4
K = 40
imgs = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, c, seed=12345 + i)).cuda() for i in range(B)]
cap = fpng_amd.max_encoded_size(w, h, c) + 64
out_sets = [[torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(B)] for _ in range(n_sets)]
enc = fpng_amd.Encoder(device=0, stream="own")
batches = [enc.make_batch(imgs, o) for o in out_sets]
for i in range(64):
    enc.submit(batches[i % n_sets], None, 0)
enc.finish(B)
for r in range(regions):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sub = []
    tickets = []
    for i in range(K):
        a = time.perf_counter()
        enc.submit(batches[i % n_sets], None, 0)
        sub.append(time.perf_counter() - a)
        tickets.append(enc.last_ticket)
    done = []
    for t in tickets[-8:]:  # (only the last eight are still known to the slot ring)
        enc.wait(t, B)
        done.append(time.perf_counter() - t0)
    enc.finish(B)
    el = time.perf_counter() - t0
    slow = sorted(range(K), key=lambda i: -sub[i])[:4]
    print(f"region {r}: {el * 1e3:.2f} ms = {B * w * h * K / el / 1e9:.1f} GP/s; slowest submit calls: " + ", ".join(f"#{i} {sub[i] * 1e3:.2f} ms" for i in slow) +
          f"; sum of submit calls {sum(sub) * 1e3:.2f} ms; last tickets done at " + " ".join(f"{d * 1e3:.1f}" for d in done), flush=True)
enc.close()
