#!/bin/bash
# Same-box comparison of several builds of the library: tools/gpu_ab_libs.sh "<bench args>" lib1.so lib2.so ...
# plus the round-1 tree (r1tree/, when present) as the fixed reference.  Boxes differ by ~10 %.
ARGS=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
p() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline'].get('phase_ms'))"; }
for i in 1 2 3; do
  [ -d r1tree ] && (cd r1tree && python bench.py --no-cpu-baseline --steps 30 --warmup 5 $ARGS 2>/dev/null | p r1)
  for lib in "$@"; do
    FPNG_AMD_LIB=$R/fpng_amd/lib/$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 $ARGS 2>/dev/null | p $lib
  done
done
