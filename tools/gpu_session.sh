#!/bin/bash
# One GPU-box session: tools/gpu_session.sh <name> -- runs gpurun_out/<name>.cmd style command lists is overkill; this is the round's
# standard opening: the -m gpu suite, smoke(), the default bench line.   usage (through gpurun): bash tools/gpu_session.sh <tag>
TAG=${1:-s}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${TAG}_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_gpu_tests.txt
tail -5 $O/${TAG}_gpu_tests.txt
timeout 300 python __graft_entry__.py smoke > $O/${TAG}_smoke.txt 2>&1; tail -2 $O/${TAG}_smoke.txt
timeout 900 python bench.py > $O/${TAG}_bench_8k.json 2> $O/${TAG}_bench_8k.err; tail -c 3000 $O/${TAG}_bench_8k.json
