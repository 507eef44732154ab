#!/bin/bash
# Round 5, encode_direct_kernel: the version with two-level look-back, 16-byte placement and a polling discipline
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
F="grep -v ^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
FPNG_AMD_DIRECT=1 timeout 280 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_kat or direct_placement" 2>&1 | $F | tail -8 | tee $O/r05_direct3_tests.txt
T() { env "$@" timeout 120 python tools/direct_timing.py $WL 2>&1 | $F | tail -1; }
L=$R/fpng_amd/lib
{ WL="7680x4320x4 8 0"
  T FPNG_AMD_DIRECT=0
  T FPNG_AMD_DIRECT=1
  T FPNG_AMD_DIRECT=1 FPNG_AMD_LIB=$L/libfpng_amd_direct_sleep4.so
  T FPNG_AMD_DIRECT=1 FPNG_AMD_LIB=$L/libfpng_amd_direct_sleep64.so
  T FPNG_AMD_DIRECT=1 FPNG_AMD_LIB=$L/libfpng_amd_direct_w8.so
  T FPNG_AMD_DIRECT=1 FPNG_AMD_PIECE_PX=1024
  T FPNG_AMD_DIRECT=1 FPNG_AMD_LANES=1
  T FPNG_AMD_DIRECT=1 FPNG_AMD_STAGGER=1
  WL="1920x1080x3 256 0"
  T FPNG_AMD_DIRECT=0
  T FPNG_AMD_DIRECT=1
  WL="512x512x3 1024 0"
  T FPNG_AMD_DIRECT=0
  T FPNG_AMD_DIRECT=1
  WL="7680x4320x4 8 1"
  T FPNG_AMD_DIRECT=0
  T FPNG_AMD_DIRECT=1
} 2>&1 | tee $O/r05_direct3_timing.txt
cat > /tmp/stats.py <<'PY'
import ctypes as C, sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd()); import fpng_amd
enc = fpng_amd.Encoder(device=0, stream="own")
for (k, w, h, c, n) in [("grad", 7680, 4320, 4, 8), ("grad", 1920, 1080, 3, 64)]:
    imgs = [torch.from_numpy(fpng_amd.synth_image(k, w, h, c, seed=12345 + i)).cuda() for i in range(n)]
    for rep in range(3):
        pngs, modes = enc.encode_tensors(imgs, 0)
    buf = (C.c_uint32 * 4)(); tot = [0, 0, 0, 0]
    for lane in (0, 1):
        if enc.lib.fpng_amd_debug_peek(enc.h, lane, buf, 4) == 0: tot = [a + b for a, b in zip(tot, list(buf))]
    print(f"{n} x {w}x{h}x{c} {k}: deferred {tot[0]}, spilled {tot[1]} of {tot[3]} chunks", flush=True)
enc.close()
PY
FPNG_AMD_DIRECT=1 timeout 100 python /tmp/stats.py 2>&1 | $F | tee -a $O/r05_direct3_timing.txt
