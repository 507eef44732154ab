#!/bin/bash
# GPU decoder: parity suite, device-resident timing, kernel stats of the timing script (arg 1 = prof)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_decode.py -x -q -m gpu ) > $O/pytest_decode.log 2>&1; tail -15 $O/pytest_decode.log
timeout 300 python tools/decode_device_timing.py 6 2>&1 | tee $O/decode_device_timing.txt
if [ "$1" = "prof" ]; then
  cd /tmp; export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode_r04 -o dec -- python $R/tools/decode_device_timing.py 10 "8K RGBA grad" > /dev/null 2>&1
  f=$(find $O/prof_decode_r04 -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | sed 's/fpng_amd::(anonymous namespace):://' | cut -c1-110 | head -16
fi
