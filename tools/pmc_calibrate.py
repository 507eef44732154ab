#!/usr/bin/env python
"""Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: streams a 1 GiB buffer with the
access widths the encoder uses.  The counter value / 2^30 gives the unit correction for that width."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import torch
from stream_probe import stream as probe  # (tools/probes/stream_probe.hip: the plain reader / writer, its own little library)
buf = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for write in (0, 1):
    for lane in (4, 16):
        for _ in range(2):
            probe(buf, write, lane)
torch.cuda.synchronize()
print("calibration done")
