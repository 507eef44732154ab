#!/bin/bash
# pipelined one-frame submissions with 2 / 3 / 4 lanes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
for WL in 4k 512 1080p 8k; do for L in 2 3 4 2 3 4; do
  FPNG_AMD_LANES=$L python bench.py --no-cpu-baseline --steps 200 --warmup 20 --batch 1 --workload $WL 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL lanes=$L', d['value'], 'MP/s', d['ms_per_step'], 'ms/frame', d['runs'], d['parity_checked'])"
done; done | tee $O/lanes_small.txt
