# round 5: the table builder's wave-parallel merge: 2-pass parity, the builder's cycle counts (timing build), the 2-pass lines and single-frame latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_train.py tests/test_gpu_real_image.py tests/test_ui_content.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -3
FPNG_AMD_LIB=$R/fpng_amd/lib/libfpng_amd_timing.so timeout 120 python tools/build_timing.py 2>&1 | grep "cycles\|inside"
line() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d.get('parity_checked'), d['roofline']['phase_ms'])"; }
timeout 200 python bench.py --no-cpu-baseline --flags 1 2>/dev/null | grep "^{" | line "8k_2pass"
timeout 200 python bench.py --no-cpu-baseline --flags 1 --workload 1080p --batch 256 2>/dev/null | grep "^{" | line "1080p_rgb_x256_2pass"
timeout 200 python bench.py --no-cpu-baseline --flags 1 --batch 1 2>/dev/null | grep "^{" | line "8k_x1_2pass"
timeout 120 python tools/latency.py 2>&1 | grep "flags=1"
