#!/bin/bash
# A/B timing of several builds of libfpng_amd.so on ONE GPU box (box-to-box clocks differ by several percent).
# usage: tools/ab_bench.sh "<bench args>" lib1.so lib2.so ...   (interleaved, 3 rounds)
args="$1"; shift
for round in 1 2 3; do
  for lib in "$@"; do
    FPNG_AMD_LIB=$PWD/$lib timeout 120 python bench.py --steps 20 --no-cpu-baseline $args 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', round(d['value']/1000,1), d['ms_per_step'], d['roofline']['phase_ms'])" "$lib"
  done
done
