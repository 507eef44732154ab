#!/usr/bin/env python
"""Summary of dec_unfilter_kernel's per-tile time stamps (diagnostic build tile_timing, FPNG_AMD_TILE_TIMES=<file>): columns item, start, rows
in the tile, own column sums known, carry known, end (100 MHz ticks)."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 1].min()
us = (a[:, 1:] - t0) / 100.0
print(f"{len(a)} tiles; kernel span {us[:, 4].max():.1f} us")
if a.shape[1] >= 9:
    for name, x in (("  start -> walks dealt (barrier)", us[:, 5] - us[:, 0]), ("  thread 0's walks", us[:, 6] - us[:, 5]), ("  -> barrier, long matches, barrier", us[:, 1] - us[:, 6])):
        print(f"  {name:32s} mean {x.mean():8.2f} us  p50 {np.percentile(x, 50):8.2f}  p90 {np.percentile(x, 90):8.2f}")
for name, x in (("fill (start -> rows there)", us[:, 1] - us[:, 0]), ("column sums", us[:, 2] - us[:, 1]), ("look-back", us[:, 3] - us[:, 2]), ("pixels out", us[:, 4] - us[:, 3]), ("whole tile", us[:, 4] - us[:, 0])):
    print(f"  {name:28s} mean {x.mean():8.2f} us  p10 {np.percentile(x, 10):8.2f}  p50 {np.percentile(x, 50):8.2f}  p90 {np.percentile(x, 90):8.2f}  max {x.max():8.2f}")
# how many tiles are alive at a time
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 4], -np.ones(len(us))], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
print(f"  tiles alive (time-weighted mean) {np.sum(alive[:-1] * dt) / max(dt.sum(), 1e-9):.1f}, max {alive.max():.0f}")
order = np.argsort(a[:, 0])
st = us[order, 0]
print("  start of item k (us): " + ", ".join(f"{k}:{st[k]:.0f}" for k in range(0, len(st), max(1, len(st) // 12))))
