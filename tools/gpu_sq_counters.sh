#!/bin/bash
# SQ-level PMC pass for bench.py (own run, kernel-trace only) -> gpurun_out/sq_<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -o sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/run_b.log 2>&1
ls $OUT $OUT/b | head
