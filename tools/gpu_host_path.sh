#!/bin/bash
# GPU-box session for the host-buffer pipelines: parity tests, then timings.  tools/gpu_host_path.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_gpu_host_pipeline.py -x -q -m gpu > $O/pytest_host_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_host_$TAG.log
(timeout 300 python tools/host_path_timing.py 2>&1 | grep -v Warn; FPNG_AMD_HOST_BANDS=4 timeout 300 python tools/host_path_timing.py 2>&1 | grep -v Warn) | tee $O/host_path_$TAG.txt
