#!/bin/bash
# Same-box A/B of two builds of the library: tools/gpu_ab_lib.sh <lib_a.so> <lib_b.so> [bench args]
# (boxes differ by ~10 % in clock/HBM behaviour: only runs on the same box are comparable)
A=$1; B=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2 3; do
  for lib in $A $B; do
    FPNG_AMD_LIB=$R/$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['roofline'].get('phase_ms'))"
  done
done
