#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_sharded_cpp.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
for i in 1 2 3; do python tools/host_state_probe.py 3 2>&1 | tail -1; done
for i in 1 2; do python bench.py --end-to-end-only cabi --workload 7680x4320x4 2>&1 | tail -1 | cut -c1-330;  python bench.py --end-to-end-only dropin --workload 7680x4320x4 2>&1 | tail -1 | cut -c1-150; done
