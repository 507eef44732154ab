#!/usr/bin/env python
"""python tools/latency_trace_summary.py <kernel_trace.csv>: anatomy of the last 100 single-frame chains of tools/latency_trace.py."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("fpng_amd::(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
chains, cur = [], []
for s, e, n in rows:
    if cur and cur[-1][2].startswith("finalize"):  # (a chain ends with finalize_kernel; the job-record upload in front is a blit kernel)
        chains.append(cur); cur = []
    cur.append((s, e, n))
chains.append(cur)
chains = [c for c in chains if c and c[-1][2].startswith("finalize")][-100:]
names = [k[2] for k in chains[-1]]
print("kernels of a chain:", ", ".join(names))
med = lambda v: sorted(v)[len(v) // 2] / 1e3
print("duration us (median): " + "  ".join(f"{n} {med([c[i][1] - c[i][0] for c in chains if len(c) == len(names)]):.1f}" for i, n in enumerate(names)))
print("gap before us (median): " + "  ".join(f"{n} {med([c[i][0] - c[i - 1][1] for c in chains if len(c) == len(names)]):.1f}" for i, n in enumerate(names) if i))
print(f"first kernel start -> last kernel end: {med([c[-1][1] - c[0][0] for c in chains]):.1f} us;  chain to chain (start to start): {med([b[0][0] - a[0][0] for a, b in zip(chains, chains[1:])]):.1f} us")
