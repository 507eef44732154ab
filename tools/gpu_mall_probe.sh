#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/mall_probe; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o mall -- python $R/tools/mall_probe.py > $O/log.txt 2>&1
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/mall_probe_summary.py $F | tee $O/summary.txt
