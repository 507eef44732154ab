#!/bin/bash
# One GPU-box session that collects everything profiles/ holds for a round (run through gpurun):
#   tools/gpu_round_all.sh <tag>
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
b() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" 2> $O/bench_${TAG}_$name.err | grep "^{" > $O/bench_${TAG}_$name.json; python - <<PY
import json
try:
    d = json.load(open("$O/bench_${TAG}_$name.json")); print("$name", d["value"], d["ms_per_step"], d.get("parity_checked"), d["roofline"].get("phase_ms", ""))
except Exception as e:
    print("$name FAILED", e)
PY
}
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_8k.json 2> $O/bench_${TAG}_8k.err; tail -c 400 $O/bench_${TAG}_8k.json; echo
b 8k_2pass --flags 1 --steps 20 --warmup 5
b 4k --workload 4k --batch 16 --steps 20 --warmup 5
b 1080p --workload 1080p --batch 256 --steps 20 --warmup 5
b 5k_4k2pass --workload 4k --batch 128 --flags 1 --steps 10 --warmup 3
b 512 --workload 512 --batch 1024 --steps 20 --warmup 5
b 8k_noise --kind noise --steps 10 --warmup 3
b 4k_blocks --workload 4k --batch 16 --kind blocks --steps 20 --warmup 5
b 16k --workload 16k --batch 1 --steps 20 --warmup 5
b 16k_rowband --mode rowband --workload 16k --steps 5 --warmup 2
timeout 200 python tools/latency.py 2>/dev/null > $O/latency_$TAG.txt; cat $O/latency_$TAG.txt
python tools/host_path_timing.py > $O/host_path_$TAG.txt 2>/dev/null; tail -5 $O/host_path_$TAG.txt
bash tools/gpu_profile_round.sh $TAG > /dev/null 2>&1
bash tools/gpu_sq_counters.sh $TAG > /dev/null 2>&1
ls $O/prof_$TAG $O/sq_$TAG | head -20
