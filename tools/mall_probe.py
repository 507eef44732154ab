#!/usr/bin/env python
"""Run under `rocprofv3 --kernel-trace`: does the 256 MiB Infinity Cache hand data from a writer kernel to a reader kernel?
Sequence per (A MiB, evict MiB): write A (16-byte lanes), read `evict` MiB of another buffer, read A back.  The kernel
trace gives each dispatch's duration; tools/mall_probe_summary.py turns them into GB/s per case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import torch
from stream_probe import stream as probe  # (tools/probes/stream_probe.hip)
big = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
a = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
plan = []
for a_mb in (16, 32, 64, 128, 256):
    for ev_mb in (0, 64, 128, 256, 512, 1024):
        for rep in range(3):
            probe(a[: a_mb << 20], 1, 16)
            if ev_mb:
                probe(big[: ev_mb << 20], 0, 16)
            probe(a[: a_mb << 20], 0, 16)
            plan.append((a_mb, ev_mb))
# write-after-write on the same lines (does a dirty line that is overwritten in cache ever reach HBM?) and read-after-read
for a_mb in (32, 64, 128):
    for rep in range(3):
        probe(a[: a_mb << 20], 1, 16)
        probe(a[: a_mb << 20], 1, 16)
        probe(a[: a_mb << 20], 0, 16)
        probe(a[: a_mb << 20], 0, 16)
torch.cuda.synchronize()
print("mall probe done", len(plan))
