#!/bin/bash
# one compact line per bench run: tools/bench_line.sh <label> <bench args...>
label="$1"; shift
timeout 180 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(sys.argv[1], '|', round(d['value'] / 1000, 1), 'GP/s', d['ms_per_step'], d['roofline']['phase_ms'], 'png_MB', round(d['config']['png_bytes_per_step_per_gpu'] / 1e6, 1))" "$label"
