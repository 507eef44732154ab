#!/usr/bin/env python
"""Per-kernel averages (per dispatch) of the counters under a directory of rocprofv3 --pmc passes."""
import collections, csv, glob, re, sys
out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else r"(dec_[a-z_]+)"
tot = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        m = re.search(pat, r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, d in agg.items():
        for c, v in d.items():
            tot[k][c] = (v / len(disp[k]), len(disp[k]))
for k, d in tot.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print(f"   {c:26s} {v/1e6:12.3f} M per dispatch ({n} dispatches)")
