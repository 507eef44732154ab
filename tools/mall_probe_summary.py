#!/usr/bin/env python
"""Summarise tools/mall_probe.py from a rocprofv3 kernel trace csv: python tools/mall_probe_summary.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "calib" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), "W" if "write" in r["Kernel_Name"] else "R", int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
seq = [(k, d) for _, k, d in rows]
i = 0
out = {}
for a_mb in (16, 32, 64, 128, 256):
    for ev_mb in (0, 64, 128, 256, 512, 1024):
        for rep in range(3):
            w = seq[i]; i += 1
            e = None
            if ev_mb:
                e = seq[i]; i += 1
            r = seq[i]; i += 1
            assert w[0] == "W" and r[0] == "R"
            out.setdefault((a_mb, ev_mb), []).append((w[1], e[1] if e else 0, r[1]))
print("A_MiB evict_MiB | write TB/s | evict-read TB/s | read-back TB/s   (best of 3)")
for (a_mb, ev_mb), v in out.items():
    bw = lambda mb, ns: mb * 1.048576 / ns * 1e3 if ns else 0
    print(f"{a_mb:5d} {ev_mb:6d} | {max(bw(a_mb, x[0]) for x in v):8.2f} | {max(bw(ev_mb, x[1]) for x in v):8.2f} | {max(bw(a_mb, x[2]) for x in v):8.2f}")
print("A_MiB | write, write again, read, read again TB/s (best of 3)")
for a_mb in (32, 64, 128):
    v = []
    for rep in range(3):
        v.append([seq[i + k][1] for k in range(4)]); i += 4
    print(f"{a_mb:5d} | " + " ".join(f"{max(a_mb * 1.048576 / x[k] * 1e3 for x in v):8.2f}" for k in range(4)))
