#!/bin/bash
# Round 5, question behind the "direct placement" design: what does a row walk cost when a row is cut into CHUNKS that fit the
# wave's LDS window (so that a chunk's stream can wait on chip for its bit offset), and how much occupancy can the walk give up for a
# larger window?  (a) the same 1.06 GB of 'grad' pixels as 8 images of other shapes -- 1280 / 1920 / 2560-pixel rows are what 6 / 4 / 3
# chunks per 7680-pixel row would cost the walker; (b) build variants: window 1024 / 1536 / 2048 dwords at 8 / 6 / 5 / 4 waves per SIMD.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
run() { # lib suffix, workload
  FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd$1.so timeout 200 python bench.py --workload $2 --steps 20 --prewarm 40 --no-cpu-baseline --decode-steps 2 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('lib%-12s %-14s %8.1f GP/s  %.4f ms/step  rows %.4f  assemble %.4f  all kernels %.4f' % ('$1' or '(product)', '$2', d['value']/1e3, d['ms_per_step'], r['phase_ms'].get('encode_rows',0), r['phase_ms'].get('assemble',0), r['all_kernels_ms']))"
}
for W in 7680x4320x4 2560x12960x4 1920x17280x4 1280x25920x4; do run "" $W; done 2>&1 | tee $O/r05_chunk_probe.txt
for V in _win1024_w6 _win1024_w4 _win1536_w5 _win2048_w4; do run $V 7680x4320x4; done 2>&1 | tee -a $O/r05_chunk_probe.txt
run "" 7680x4320x4 2>&1 | tee -a $O/r05_chunk_probe.txt
