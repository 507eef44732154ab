"""Host time of one fpng_amd_encoder_submit() call against the step time (round 5): is a submission of B images host-bound?
    python tools/submit_cost.py [w h c B] ...   (default: the four bench shapes);  FPNG_AMD_LANES as in bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fpng_amd
shapes = [(512, 512, 3, 1024), (256, 256, 3, 2048), (1920, 1080, 3, 256), (7680, 4320, 4, 8)]
if len(sys.argv) > 1:
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
for (w, h, c, B) in shapes:
    imgs = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, c, seed=12345 + (i % 16))).cuda() for i in range(B)]
    cap = fpng_amd.max_encoded_size(w, h, c) + 64
    n_sets = 4
    outs = [[torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(B)] for _ in range(n_sets)]
    enc = fpng_amd.Encoder(device=0, stream="own")
    batches = [enc.make_batch(imgs, o) for o in outs]
    K = max(20, 160 // B)
    for i in range(K): enc.submit(batches[i % n_sets], None, 0)
    enc.finish(B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): enc.submit(batches[i % n_sets], None, 0)
    t1 = time.perf_counter(); enc.finish(B); torch.cuda.synchronize(); t2 = time.perf_counter()
    enc.finish(B); torch.cuda.synchronize(); t3 = time.perf_counter()
    for i in range(6): enc.submit(batches[i % n_sets], None, 0)
    t4 = time.perf_counter(); enc.finish(B); torch.cuda.synchronize()
    print(f"   six submits from idle (no slot to wait for): {(t4 - t3) / 6 * 1e3:.3f} ms of host time each")
    print(f"{B} x {w}x{h}x{c} lanes={os.environ.get('FPNG_AMD_LANES', 'default')}: host time per submit {(t1 - t0) / K * 1e3:.3f} ms, per step {(t2 - t0) / K * 1e3:.3f} ms, "
          f"{B * w * h / ((t2 - t0) / K) / 1e9:.1f} GP/s", flush=True)
    enc.close()
