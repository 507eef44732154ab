#!/bin/bash
# The host paths in fresh processes, several times each: do the two copy directions overlap every time?
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/copy_streams.txt; : > $O
FPNG_AMD_TRACE=1 python tools/host_batch_order_probe.py A 2>&1 | grep "copy streams" >> $O
for rep in 1 2 3 4; do
  for m in A B; do python tools/host_batch_order_probe.py $m 2>&1 | tail -1 >> $O; done
  echo -n "bench e2e: " >> $O; python bench.py --end-to-end-only cabi --workload 7680x4320x4 2>&1 | tail -1 | cut -c1-120 >> $O
  echo -n "dropin: " >> $O; python bench.py --end-to-end-only dropin --workload 7680x4320x4 2>&1 | tail -1 | cut -c1-120 >> $O
  python tools/host_state_probe.py 3 2>&1 | tail -1 >> $O
done
cat $O
