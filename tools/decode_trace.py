#!/usr/bin/env python
"""FPNG_AMD_TRACE=1 timeline of one fpng_amd_decode_batch call (4 x 8K RGBA grad, then 16 x 4K)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, fpng_amd
enc = fpng_amd.Encoder(device=0)
for name, (w, h, n) in (("8K x 4", (7680, 4320, 4)), ("4K x 16", (3840, 2160, 16))):
    ts = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, 4, seed=12345 + i)).cuda() for i in range(n)]
    pngs, _ = enc.encode_tensors(ts, 0)
    dims = [(w, h)] * n
    for rep in range(3):
        os.environ["FPNG_AMD_TRACE"] = "1"
        print(f"== {name}, call {rep}", file=sys.stderr, flush=True)
        torch.cuda.synchronize(); t0 = time.perf_counter(); enc.decode_batch(pngs, 4, dims); torch.cuda.synchronize()
        print(f"   {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
