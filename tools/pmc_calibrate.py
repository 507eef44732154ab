#!/usr/bin/env python
"""Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`: streams a 1 GiB buffer with the
access widths the encoder uses.  The counter value / 2^30 gives the unit correction for that width."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpng_amd
enc = fpng_amd.Encoder(device=0, stream="own")
buf = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for write in (0, 1):
    for lane in (4, 16):
        for _ in range(2):
            enc.calibration_stream(buf, write, lane)
print("calibration done")
