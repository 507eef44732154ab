#!/bin/bash
# Round-end collection: default bench line (cpu baseline, end_to_end), other workloads, row-band lines, latency.  tools/gpu_round_final.sh <tag>
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out
timeout 600 python bench.py > $O/bench_${TAG}_8k_full.json 2> $O/bench_${TAG}_8k_full.err; tail -c 1500 $O/bench_${TAG}_8k_full.json; echo
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep "^{" > $O/bench_${TAG}_$name.json; python -c "import json; d=json.load(open('$O/bench_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('runs'), d['parity_checked'], d['roofline'].get('phase_ms',''))"; }
b 8k_2pass --flags 1 --steps 20 --warmup 5
b 4k --workload 4k --batch 16 --steps 20 --warmup 5
b 4k_blocks --workload 4k --batch 16 --kind blocks --steps 20 --warmup 5
b 1080p --workload 1080p --batch 256 --steps 20 --warmup 5
b 5k_4k2pass --workload 4k --batch 128 --flags 1 --steps 10 --warmup 3
b 512 --workload 512 --batch 1024 --steps 20 --warmup 5
b 16k --workload 16k --batch 1 --steps 20 --warmup 5
b 16k_rowband --mode rowband --workload 16k --steps 10 --warmup 3
b 16k_rowband_2pass --mode rowband --workload 16k --steps 10 --warmup 3 --flags 1
timeout 200 python tools/latency.py 2>/dev/null | tee $O/latency_$TAG.txt
