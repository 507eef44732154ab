#!/bin/bash
# N alternating runs per library, ms_per_step only: tools/gpu_ab_many.sh N "<bench args>" lib1.so lib2.so ...
N=$1; ARGS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for lib in "$@"; do : > /tmp/ab_$lib.txt; done
for i in $(seq $N); do
  for lib in "$@"; do
    FPNG_AMD_LIB=$R/fpng_amd/lib/$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> /tmp/ab_$lib.txt
  done
done
for lib in "$@"; do echo "$lib [$ARGS]: $(sort -n /tmp/ab_$lib.txt | tr '\n' ' ')"; done
