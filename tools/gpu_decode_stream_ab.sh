#!/bin/bash
# fpng::fpng_decode_memory on large files (the streamed fpng_amd_decode_host): the Up filter's undoing on its own stream and the ramp
# of small first pieces, each switched off and on (same box); first the tests that go through that path
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_dropin_decode.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    echo "== unf_stream=$1 ramp=$2"
    FPNG_AMD_DECODE_UNF_STREAM=$1 FPNG_AMD_DECODE_RAMP=$2 timeout 200 python tools/dropin_decode_timing.py 2>/dev/null | grep "8K\|4K\|2748"
  done
done | tee $O/decode_stream_ab.txt
