#!/bin/bash
# A/B: copy streams with a priority of their own (their own hardware queues) vs. ordinary streams, each in fresh processes.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/copy_priority.txt; : > $O
for rep in 1 2; do
for p in 1 0; do
  for m in A B; do
    echo -n "priority=$p " >> $O; FPNG_AMD_COPY_PRIORITY=$p python tools/host_batch_order_probe.py $m 2>&1 | tail -1 >> $O
  done
  echo -n "priority=$p bench e2e: " >> $O; FPNG_AMD_COPY_PRIORITY=$p python bench.py --end-to-end-only cabi --workload 7680x4320x4 2>&1 | tail -1 | cut -c1-120 >> $O
  echo -n "priority=$p dropin: " >> $O; FPNG_AMD_COPY_PRIORITY=$p python bench.py --end-to-end-only dropin --workload 7680x4320x4 2>&1 | tail -1 | cut -c1-120 >> $O
done
done
cat $O
