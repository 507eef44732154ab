R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/clk; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O -o sq -- python $R/tools/decode_device_timing.py 2 "8K RGBA grad" > $O/run.log 2>&1
python3 - <<PY
import csv, glob, collections, re
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob("$O/*counter_collection.csv")[0])):
    m = re.search(r"(dec_[a-z_]+|encode_rows_kernel)", r["Kernel_Name"])
    if m: cnt[(r["Dispatch_Id"], m.group(1))][r["Counter_Name"]] = float(r["Counter_Value"])
dur = {}
for r in csv.DictReader(open(glob.glob("$O/*kernel_trace.csv")[0])):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for (d, k), c in list(cnt.items())[-12:]:
    ns = dur.get(d, 0)
    print(k, "dur us", ns / 1e3, {n: v for n, v in c.items()}, "GUI_ACTIVE/ns =", round(c.get("GRBM_GUI_ACTIVE", 0) / max(ns, 1), 3))
PY
