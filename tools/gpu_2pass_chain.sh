#!/bin/bash
# 2-pass chain without the histogram memset / the walked marker for lone frames: parity + latency + one 2-pass bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_train.py -x -q -m gpu ) > $O/pytest_2pc.log 2>&1; tail -4 $O/pytest_2pc.log
python tools/latency.py 2>/dev/null | tee $O/latency_2pc.txt
timeout 300 python bench.py --no-cpu-baseline --flags 1 --steps 30 --warmup 5 2>/dev/null | grep "^{" | tee $O/bench_2pc.json | cut -c1-420
