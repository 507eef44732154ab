#!/usr/bin/env python
"""Summary of dec_sync_kernel<false>'s per-workgroup time stamps (diagnostic build sync_timing, FPNG_AMD_SYNC_TIMES=<file>): columns workgroup,
start, bits staged, thread 0's wave decoded, all waves decoded (first barrier of the corrections), corrections done, end (100 MHz ticks)."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 1].min()
us = (a[:, 1:] - t0) / 100.0
print(f"{len(a)} workgroups; kernel span {us[:, 5].max():.1f} us")
for name, x in (("staging (table + bits into LDS)", us[:, 1] - us[:, 0]), ("wave 0: lead-in + its tokens", us[:, 2] - us[:, 1]), ("... until the last wave is there", us[:, 3] - us[:, 2]),
                ("corrections", us[:, 4] - us[:, 3]), ("results out, block record", us[:, 5] - us[:, 4]), ("whole workgroup", us[:, 5] - us[:, 0])):
    print(f"  {name:34s} mean {x.mean():8.2f} us  p10 {np.percentile(x, 10):8.2f}  p50 {np.percentile(x, 50):8.2f}  p90 {np.percentile(x, 90):8.2f}  max {x.max():8.2f}")
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 5], -np.ones(len(us))], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0])
print(f"  workgroups alive (time-weighted mean) {np.sum(alive[:-1] * dt) / max(dt.sum(), 1e-9):.1f}, max {alive.max():.0f}")
