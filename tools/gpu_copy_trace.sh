#!/bin/bash
# Which engine moves the host path's bytes, and do the two directions overlap?  rocprofv3 memory-copy + kernel trace of the
# order probe in both orders and of bench.py's end-to-end leg.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/copytrace; rm -rf $O; mkdir -p $O
for m in A B; do
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/$m -- python tools/host_batch_order_probe.py $m > $O/$m.log 2>&1
  tail -1 $O/$m.log
done
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/E -- python bench.py --end-to-end-only cabi --workload 7680x4320x4 > $O/E.log 2>&1
tail -1 $O/E.log
find $O -name "*.csv" | xargs ls -la
# keep only the copy traces and kernel traces
find $O -type f ! -name "*memory_copy_trace.csv" ! -name "*kernel_trace.csv" ! -name "*.log" -delete
