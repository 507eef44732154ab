#!/bin/bash
# kernel trace of the pipelined default bench (two lanes): how do the lanes' kernels overlap in steady state?
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/overlap; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python bench.py --no-cpu-baseline --steps 12 --warmup 2 --regions 1 > $O/run.log 2>&1
tail -1 $O/run.log | cut -c1-200
ls $O
