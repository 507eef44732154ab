#!/bin/bash
# final-tree collection: the -m gpu suite into a log, then the bench lines of the other workloads (no CPU baseline)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_final.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_final.log | tail -2
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | grep "^{" > $O/bench_final_$name.json; python -c "import json; d=json.load(open('$O/bench_final_$name.json')); print('$name', d['value'], d['ms_per_step'], d['runs'], d['parity_checked'], d['roofline']['phase_ms'])"; }
b 8k_2pass --flags 1
b 1080p --workload 1080p --batch 256
b 512 --workload 512 --batch 1024
b 4k --workload 4k --batch 16
b 8k_noise --kind noise
b 16k --workload 16k --batch 1
