#!/bin/bash
# A/B of the submission schedules (FPNG_AMD_SCHED=stages|lanes) on the bench workloads
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
line() { python - "$@" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); print(name, d["value"], d["ms_per_step"], d.get("parity_checked"))
except Exception as e:
    print(name, "FAILED", e)
PY
}
for sched in stages lanes; do
  for spec in "8k:--steps 30 --warmup 5" "8k_2pass:--flags 1 --steps 20 --warmup 5" "4k:--workload 4k --batch 16 --steps 30 --warmup 5" "1080p:--workload 1080p --batch 256 --steps 20 --warmup 5" "512:--workload 512 --batch 1024 --steps 20 --warmup 5" "4k_b1:--workload 4k --batch 1 --steps 200 --warmup 20" "8k_noise:--kind noise --steps 10 --warmup 3"; do
    name=${spec%%:*}; args=${spec#*:}
    FPNG_AMD_SCHED=$sched timeout 300 python bench.py --no-cpu-baseline $args > $O/ab_${sched}_$name.json 2> $O/ab_${sched}_$name.err
    line "$sched/$name" $O/ab_${sched}_$name.json
  done
done
timeout 200 python tools/latency.py 2>/dev/null | head -8
