#!/bin/bash
# Round 5: first GPU session of the direct-placement encoder (encode_direct_kernel): parity first, then statistics and timings.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
F="grep -v ^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
timeout 280 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_kat or direct_placement or smoke" 2>&1 | $F | tail -15 | tee $O/r05_direct1_tests.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r05_direct1_tests.txt
cat > /tmp/stats.py <<'PY'
import ctypes as C, sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd()); import fpng_amd
enc = fpng_amd.Encoder(device=0, stream="own")
for (k, w, h, c, n) in [("grad", 7680, 4320, 4, 8), ("grad", 1920, 1080, 3, 64), ("noise", 3840, 2160, 4, 4), ("blocks", 3840, 2160, 4, 8)]:
    imgs = [torch.from_numpy(fpng_amd.synth_image(k, w, h, c, seed=12345 + i)).cuda() for i in range(n)]
    for flags in (0, 1):
        for rep in range(3):
            pngs, modes = enc.encode_tensors(imgs, flags)
        buf = (C.c_uint32 * 4)()
        tot = [0, 0, 0, 0]
        for lane in (0, 1):
            if enc.lib.fpng_amd_debug_peek(enc.h, lane, buf, 4) == 0:
                tot = [a + b for a, b in zip(tot, list(buf))]
        print(f"{n} x {w}x{h}x{c} {k} flags {flags}: deferred {tot[0]}, spilled {tot[1]} of {tot[3]} chunks (both lanes' last submissions), modes {sorted(set(modes))}", flush=True)
enc.close()
PY
timeout 200 python /tmp/stats.py 2>&1 | $F | tee $O/r05_direct1_stats.txt
run() { # label, env..., -- bench args
  local label=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --prewarm 40 --no-cpu-baseline --decode-steps 2 $BARGS 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('%-34s %8.1f GP/s  %.4f ms/step  parity %s  %s' % ('$label', d['value']/1e3, d['ms_per_step'], d['parity_checked'], ' '.join('%s %.4f' % (k, v) for k, v in r['phase_ms'].items() if v > 0.002)))"
}
BARGS="" 
{ run "two-kernel chain (DIRECT=0)" FPNG_AMD_DIRECT=0
  run "direct, piece 1536 (product)" FPNG_AMD_DIRECT=1
  run "direct, piece 1280" FPNG_AMD_PIECE_PX=1280
  run "direct, piece 1024" FPNG_AMD_PIECE_PX=1024
  run "direct, piece 2048 (spills)" FPNG_AMD_PIECE_PX=2048
  run "direct win1280 piece 2048" FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_direct_win1280.so FPNG_AMD_PIECE_PX=2048
  run "direct win1280 piece 1536" FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_direct_win1280.so
  run "direct 8 waves piece 1536" FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_direct_w8.so
  run "direct, one lane" FPNG_AMD_LANES=1
  run "direct, staggered walks" FPNG_AMD_STAGGER=1
  run "two-kernel chain (DIRECT=0)" FPNG_AMD_DIRECT=0
  run "direct, piece 1536 (product)" FPNG_AMD_DIRECT=1
  BARGS="--flags 1"
  run "2-pass two-kernel chain" FPNG_AMD_DIRECT=0
  run "2-pass direct" FPNG_AMD_DIRECT=1
  BARGS="--workload 1080p --batch 256"
  run "1080p RGB x256 two-kernel" FPNG_AMD_DIRECT=0
  run "1080p RGB x256 direct" FPNG_AMD_DIRECT=1
  BARGS="--workload 512 --batch 1024"
  run "512 RGB x1024 two-kernel" FPNG_AMD_DIRECT=0
  run "512 RGB x1024 direct" FPNG_AMD_DIRECT=1
} 2>&1 | tee $O/r05_direct1_bench.txt
