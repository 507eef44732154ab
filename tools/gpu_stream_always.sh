#!/bin/bash
# With the two directions pinned to two copy engines: does streaming still need buffers that were seen before, small pages, ...?
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/stream_always.txt; : > $O
for i in 1 2; do
  for m in HH H4 4H 44; do python tools/host_page_size_probe.py $m 2>&1 | tail -1 >> $O; done
  for m in HH 44; do echo -n "forced bands: " >> $O; FPNG_AMD_HOST_BANDS=8 python tools/host_page_size_probe.py $m 2>&1 | tail -1 >> $O; done
  echo -n "forced bands: " >> $O; FPNG_AMD_HOST_BANDS=8 python tools/host_batch_order_probe.py B 2>&1 | tail -1 >> $O
done
cat $O
