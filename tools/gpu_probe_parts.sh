#!/bin/bash
# Zero-code probe (round 3): does the Infinity Cache hand the local streams from encode_rows to assemble when a
# submission is small?  Per-kernel event times (one lane, serial) and pipelined throughput for B = 1/2/4/8 8K frames per
# submission, 2/3/4 lanes, with and without the non-temporal hint on the local-stream stores.
#   tools/gpu_probe_parts.sh <tag>
TAG=${1:-r03a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/probe_$TAG; mkdir -p $O
p() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline'].get('phase_ms'))
except Exception as e: print('$1 FAILED', e)"; }
for LIB in libfpng_amd.so libfpng_amd_nont.so; do
  for B in 1 2 4 8; do
    for L in 2 3 4; do
      [ $B -gt 2 ] && [ $L -gt 2 ] && continue
      FPNG_AMD_LIB=$R/fpng_amd/lib/$LIB FPNG_AMD_LANES=$L timeout 200 python bench.py --no-cpu-baseline --batch $B --steps $((240/B)) --warmup 5 --prewarm $((480/B)) 2>$O/err.txt | tee $O/${LIB}_b${B}_l${L}.json | p "$LIB B=$B lanes=$L"
    done
  done
done 2>&1 | tee $O/summary.txt
