#!/bin/bash
# SQ instruction counts of the encode kernels for rows of different widths (what do a row's start-up, its tail windows and the
# 3-channel layout cost?): one rocprofv3 --pmc pass of bench.py per workload "WxHxC:batch"
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/sqw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  wl=${spec%%:*}; b=${spec##*:}
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/$wl -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $wl --batch $b > $O/$wl.log 2>&1
  python - "$O/$wl/sq_counter_collection.csv" "$wl" "$b" <<'PY'
import csv, sys, collections
path, wl, b = sys.argv[1], sys.argv[2], int(sys.argv[3])
w, h, c = (int(v) for v in wl.split("x"))
units = w * h * b / 64
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "encode_rows" not in k: continue
    m = {cn: sum(v) / len(v) for cn, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(f"{wl} x {b} {k}: per 64 px VALU {m['SQ_INSTS_VALU']/units:.2f} SALU {m['SQ_INSTS_SALU']/units:.2f} LDS {m['SQ_INSTS_LDS']/units:.2f}; per row VALU {m['SQ_INSTS_VALU']/(h*b):.0f} SALU {m['SQ_INSTS_SALU']/(h*b):.0f}; parked {100*m['SQ_WAIT_ANY']/wc:.0f} % stalled {100*m['SQ_WAIT_INST_ANY']/wc:.0f} % VALU busy {100*m['SQ_ACTIVE_INST_VALU']/wc:.1f} %")
PY
done
