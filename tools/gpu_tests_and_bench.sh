#!/bin/bash
# GPU-box session: the -m gpu suite, then the default bench line.  tools/gpu_tests_and_bench.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_$TAG.log
timeout 600 python bench.py > $O/bench_${TAG}_8k.json 2> $O/bench_${TAG}_8k.err; tail -c 2500 $O/bench_${TAG}_8k.json
