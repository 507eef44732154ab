#!/usr/bin/env python
"""Unpipelined single-frame submissions for `rocprofv3 --kernel-trace`: tools/latency_trace.py <w> <h> <c> [flags]
(tools/latency_trace_summary.py turns the kernel trace into the chain's anatomy: kernel durations and the gaps between them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, fpng_amd
w, h, c = (int(v) for v in sys.argv[1:4])
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
enc = fpng_amd.Encoder(device=0, stream="own")
img = torch.from_numpy(fpng_amd.synth_image("grad", w, h, c)).cuda()
out = torch.empty(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device="cuda")
batch = enc.make_batch([img], [out])
for _ in range(30):
    enc.submit(batch, None, flags); enc.finish(1)
ts = []
for _ in range(100):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); enc.submit(batch, None, flags); t1 = time.perf_counter(); enc.finish(1); ts.append((t1 - t0, time.perf_counter() - t0))
ts.sort(key=lambda v: v[1])
print(f"{w}x{h}x{c} flags={flags}: submit() returns after {ts[50][0]*1e6:.1f} us, finish() after {ts[50][1]*1e6:.1f} us (median of 100)")
