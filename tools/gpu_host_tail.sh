#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
python -m pytest tests/test_gpu_host_pipeline.py -x -q > gpurun_out/pytest_host_tail.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_host_tail.log
for i in 1 2; do python tools/host_page_size_probe.py 44 2>&1 | tail -1; done
FPNG_AMD_TRACE=1 python tools/host_page_size_probe.py 44 2>&1 | grep -E "band|joined" | tail -11
python - <<'PY'
import time, numpy as np, fpng_amd
enc = fpng_amd.Encoder(device=0, stream="own")
for (w, h, c) in [(3840, 2160, 4), (5120, 2880, 4), (7680, 4320, 3)]:
    img = fpng_amd.synth_image("grad", w, h, c); out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); enc.encode_host_into(img, w, h, c, out, 0); ts.append(round((time.perf_counter() - t0) * 1e3, 3))
    print(w, h, c, ts, enc.last_host_bands())
PY
