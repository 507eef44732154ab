#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/lat_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "512 512 3" "3840 2160 4" "7680 4320 4"; do
  tag=$(echo $cfg | tr ' ' x)
  python $R/tools/latency_trace.py $cfg 2>/dev/null | tail -1
  rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o t -- python $R/tools/latency_trace.py $cfg > /dev/null 2>&1
  python $R/tools/latency_trace_summary.py $(find $O/$tag -name "*kernel_trace.csv" | head -1)
done 2>&1 | tee $O/summary.txt
