#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_trace.csv) per kernel.

    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db > profiles/r01_kernel_stats.txt
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(n, e - s) for n, s, e in cur.execute("select name, start, end from kernels")]


def from_csv(path):
    out = []
    with open(path) as f:
        for row in csv.DictReader(f):
            out.append((row["Kernel_Name"], int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    return out


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = defaultdict(list)
    for name, ns in rows:
        agg[name].append(ns)
    total = sum(sum(v) for v in agg.values())
    print(f"# source: {path}")
    print(f"{'kernel':<72} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10} {'pct':>6}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name.replace("fpng_amd::(anonymous namespace)::", "").split("(")[0]
        print(f"{short:<72} {len(v):>6} {sum(v)/len(v)/1e3:>10.1f} {min(v)/1e3:>10.1f} {max(v)/1e3:>10.1f} {sum(v)/1e6:>10.3f} {100*sum(v)/total:>6.1f}")


if __name__ == "__main__":
    main()
