#!/bin/bash
# parity suite + A/B of FPNG_AMD_ALWAYS_ORDER (ordering packets in front of every chain vs only when needed) + one default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_image.py -x -q -m gpu ) > $O/pytest_order.log 2>&1; tail -4 $O/pytest_order.log
for v in 1 0 1 0; do echo "== FPNG_AMD_ALWAYS_ORDER=$v"; FPNG_AMD_ALWAYS_ORDER=$v python tools/latency.py 2>/dev/null; done | tee $O/latency_order.txt
timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | grep "^{" | tee $O/bench_order.json | cut -c1-400
