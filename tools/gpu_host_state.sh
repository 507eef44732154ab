#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/host_state.txt; : > $O
for i in 1 2 3 4 5 6 7 8 9 10; do python tools/host_state_probe.py 3 2>&1 | tail -1 >> $O; done
cat $O
