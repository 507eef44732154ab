#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --kernel-trace --stats output directory (kernel_stats.csv + kernel_trace.csv): averages, and the
last decode step's launch sequence with start offsets."""
import csv, glob, re, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "dec_"
rows = list(csv.DictReader(open(glob.glob(d + "/*kernel_stats.csv")[0])))
for r in rows[:14]:
    name = re.sub(r"fpng_amd::\(anonymous namespace\)::", "", r["Name"]).split("(")[0]
    print(f"{name[:44]:44s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
rows = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
dec = [r for r in rows if pat in r["Kernel_Name"]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
last = dec[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = re.sub(r"fpng_amd::\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
    print(f"  {name[:28]:28s} +{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us  grid {r['Grid_Size_X']:>9s} lds {r['LDS_Block_Size']:>6s} vgpr {r['VGPR_Count']:>3s} sgpr {r['SGPR_Count']:>3s}")
