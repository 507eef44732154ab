#!/bin/bash
# PMC passes for the device-resident decode (tools/decode_device_timing.py): SQ counters in three passes, FETCH_SIZE and WRITE_SIZE
# in their own (MI355X_MICROARCH.md) -> gpurun_out/sq_decode_<tag>/ ; tools/pmc_summary.py prints per-kernel averages
TAG=${1:-r04}; CASE=${2:-8K RGBA grad}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_decode_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { d=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$d -o sq -- python $R/tools/decode_device_timing.py 2 "$CASE" > $OUT/run_$d.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES
run c SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_IFETCH SQ_ACTIVE_INST_VMEM
run f FETCH_SIZE
run w WRITE_SIZE
python3 $R/tools/pmc_summary.py $OUT
