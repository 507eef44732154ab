#!/bin/bash
# full -m gpu suite + a few bench lines + single-frame latency: tools/gpu_quick.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_$TAG.log
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep "^{" > $O/bench_${TAG}_$name.json; python -c "import json; d=json.load(open('$O/bench_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], d['runs'], d['parity_checked'], d['roofline']['phase_ms'])"; }
b 8k --steps 20 --warmup 5
b 8k_noise --kind noise --steps 10 --warmup 3
b 8k_stored --flags 2 --steps 10 --warmup 3
for cfg in "512 512 3" "3840 2160 4" "7680 4320 4"; do python tools/latency_trace.py $cfg 2>/dev/null | tail -1; done
