#!/bin/bash
# round 6 decoder sessions: the decode tests, then timings.  usage (through gpurun): bash tools/gpu_r06_decode.sh <tag> [quick]
TAG=${1:-d}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_decode.py -x -q > $O/${TAG}_decode_tests.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_decode_tests.txt
tail -15 $O/${TAG}_decode_tests.txt
FPNG_TIMING_PHASES=1 timeout 600 python tools/decode_device_timing.py 6 > $O/${TAG}_decode_timing.txt 2>&1; cat $O/${TAG}_decode_timing.txt | cut -c1-230
