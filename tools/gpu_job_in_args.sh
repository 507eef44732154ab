#!/bin/bash
# A/B of FPNG_AMD_JOB_IN_ARGS (job record of a one-image submission in the kernel arguments vs uploaded) + the parity suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) > $O/pytest_jia.log 2>&1; tail -4 $O/pytest_jia.log
for v in 0 1 0 1; do echo "== FPNG_AMD_JOB_IN_ARGS=$v"; FPNG_AMD_JOB_IN_ARGS=$v python tools/latency.py 2>/dev/null; done | tee $O/latency_jia.txt
