#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes (FETCH_SIZE / WRITE_SIZE in their
# own runs, as MI355X_MICROARCH.md prescribes) for bench.py and for the calibration streams.
# usage: tools/gpu_profile_round.sh <tag> [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# one lane: kernels of consecutive submissions do not overlap, so the per-kernel durations are the ones bench.py
# measures with events (its profiling steps also use one lane)
FPNG_AMD_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_$C.log 2>&1
  [ -z "$SKIP_CAL" ] && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/cal_$C -o cal -- python $R/tools/pmc_calibrate.py > $OUT/cal_$C.log 2>&1
done
find $OUT -name "*.csv" | head -40
