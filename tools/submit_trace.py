import sys, time, torch
sys.path.insert(0, ".")
import fpng_amd
w,h,c,B = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (1920,1080,3,256)))
imgs = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, c, seed=i)).cuda() for i in range(B)]
cap = fpng_amd.max_encoded_size(w, h, c) + 64
outs = [[torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(B)] for _ in range(4)]
enc = fpng_amd.Encoder(device=0, stream="own")
bs = [enc.make_batch(imgs, o) for o in outs]
for i in range(4):
    enc.submit(bs[i & 3]); enc.finish(B)
torch.cuda.synchronize()
for rep in range(4):
    ts = []
    t00 = time.perf_counter()
    for i in range(24):
        t0 = time.perf_counter(); enc.submit(bs[i & 3]); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); enc.finish(B); tf = (time.perf_counter() - t0) * 1e3
    tot = (time.perf_counter() - t00) * 1e3
    print(f"rep {rep}: total {tot:.1f} ms ({tot/24:.3f}/step) finish {tf:.2f}; submit ms:", " ".join(f"{x:.2f}" for x in ts[:12]))
