#!/bin/bash
# rows per encode_rows workgroup (FPNG_ROW_WAVES variants) on the short-row workloads: tools/gpu_rows_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
for args in "--workload 512 --batch 1024" "--workload 1080p --batch 256"; do for i in 1 2; do for lib in libfpng_amd.so libfpng_amd_rows8.so libfpng_amd_rows2.so; do
  FPNG_AMD_LIB=$R/fpng_amd/lib/$lib python bench.py --no-cpu-baseline --steps 30 --warmup 5 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args', '$lib', d['value'], d['ms_per_step'], d['parity_checked'], d['roofline']['phase_ms'])"
done; done; done | tee $O/rows_ab.txt
