#!/bin/bash
# same-box A/B of library builds on single-frame latency and the default bench: tools/gpu_ab_latency.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
for i in 1 2; do for lib in "$@"; do echo "== $lib"; FPNG_AMD_LIB=$R/fpng_amd/lib/$lib python tools/latency.py 2>/dev/null | grep "flags=0"; done; done | tee $O/ab_latency.txt
for i in 1 2; do for lib in "$@"; do FPNG_AMD_LIB=$R/fpng_amd/lib/$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['runs'], d['parity_checked'], d['roofline']['phase_ms'])"; done; done | tee $O/ab_bench.txt
