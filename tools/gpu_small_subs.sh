#!/bin/bash
# pipelined submissions of ONE frame each (BASELINE config 2 as literally stated, config 1's size): ordering packets old / new
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 200 --warmup 20 --batch 1 --workload $WL 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL', '$tag', d['value'], 'MP/s', d['ms_per_step'], 'ms/frame', d['runs'], d['parity_checked'])"; }
for WL in 4k 512 1080p 8k; do
  for i in 1 2; do
    run "old: packets always, job uploaded" FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_base.so FPNG_AMD_ALWAYS_ORDER=1 FPNG_AMD_JOB_IN_ARGS=0
    run "base: packets when busy" FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_base.so
    run "new: no same-stream wait" FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd.so
  done
done | tee $O/small_subs.txt
