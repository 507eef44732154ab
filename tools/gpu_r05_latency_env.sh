# round 5: single-frame latency against runtime knobs of the HIP runtime (kernel arguments in device memory, active wait)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== default"; timeout 120 python tools/latency.py 2>&1 | grep flags
echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 timeout 120 python tools/latency.py 2>&1 | grep flags
echo "== FPNG_AMD_KEEP_HW_QUEUES=1"; FPNG_AMD_KEEP_HW_QUEUES=1 timeout 120 python tools/latency.py 2>&1 | grep flags
echo "== GPU_MAX_HW_QUEUES=2"; GPU_MAX_HW_QUEUES=2 timeout 120 python tools/latency.py 2>&1 | grep flags
