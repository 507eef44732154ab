#!/bin/bash
# Round 5, last session: the whole -m gpu suite + smoke() + the five bench lines on the final tree (kernel stats / PMC traffic: tools/gpu_round5_final.sh, same kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 | tee $O/r05_last_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/r05_last_tests.txt
timeout 600 python bench.py > $O/bench_r05_8k.json 2> $O/bench_r05_8k.err; tail -c 300 $O/bench_r05_8k.json; echo
timeout 300 python bench.py --no-cpu-baseline --flags 1 2>/dev/null | grep "^{" > $O/bench_r05_8k_2pass.json
timeout 300 python bench.py --mode decode 2>/dev/null | grep "^{" > $O/bench_r05_decode_8k.json
timeout 300 python bench.py --no-cpu-baseline --workload 1080p --batch 256 2>/dev/null | grep "^{" > $O/bench_r05_1080p.json
timeout 300 python bench.py --no-cpu-baseline --workload 512 --batch 1024 2>/dev/null | grep "^{" > $O/bench_r05_512.json
for f in 8k 8k_2pass decode_8k 1080p 512; do python -c "import json; d=json.load(open('$O/bench_r05_$f.json')); print('$f', d['value'], d['ms_per_step'], d['runs'], d.get('parity_checked'), d['roofline'].get('frac'), d['roofline'].get('phase_ms'))"; done
FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_timing.so timeout 120 python tools/build_timing.py 2>&1 | grep "cycles\|inside" | tee $O/r05_build_timing.txt
timeout 120 python tools/latency.py 2>&1 | grep "flags=" | tee $O/r05_latency_last.txt
timeout 200 python bench.py --no-cpu-baseline --flags 1 --batch 1 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('8k_x1_2pass', d['value'], d['ms_per_step'], d['roofline']['phase_ms'])"
