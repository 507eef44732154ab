#!/bin/bash
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sharded_cpp.py -x -q -m gpu > $O/pytest_sharded_$TAG.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_sharded_$TAG.log
for F in 0 1; do
timeout 600 python bench.py --mode rowband --workload 16k --steps 10 --warmup 3 --flags $F 2>$O/rowband_$TAG.err | grep "^{" > $O/bench_${TAG}_16k_rowband_f$F.json; tail -c 900 $O/bench_${TAG}_16k_rowband_f$F.json; echo; tail -3 $O/rowband_$TAG.err
done
