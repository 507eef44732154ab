#!/bin/bash
# Round 5, the very last session (after the table builder's limiter): the whole -m gpu suite + smoke(), the default and the 2-pass bench line, the builder's cycle counts, 2-pass latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 | tee $O/r05_last_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r05_last_tests.txt
FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_timing.so timeout 120 python tools/build_timing.py 2>&1 | grep "cycles\|inside" | tee $O/r05_build_timing.txt
timeout 300 python bench.py --no-cpu-baseline --flags 1 2>/dev/null | grep "^{" > $O/bench_r05_8k_2pass.json
timeout 120 python tools/latency.py 2>&1 | grep "flags=" | tee $O/r05_latency_last.txt
timeout 600 python bench.py > $O/bench_r05_8k.json 2> $O/bench_r05_8k.err
for f in 8k 8k_2pass; do python -c "import json; d=json.load(open('$O/bench_r05_$f.json')); print('$f', d['value'], d['ms_per_step'], d['runs'], d.get('parity_checked'), d['roofline'].get('frac'), d['roofline'].get('phase_ms'))"; done
