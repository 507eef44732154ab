#!/bin/bash
# unpipelined latency of several builds on one box: tools/gpu_latency_ab.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2; do for lib in "$@"; do echo "== $lib"; FPNG_AMD_LIB=$R/fpng_amd/lib/$lib timeout 200 python tools/latency.py 2>/dev/null | grep "flags=0"; done; done
