"""Second encoder of a process: slow because the first one was DESTROYED (its device memory recycled), or because it is second?
mode keep: enc0 stays alive while enc1 is measured; mode close: enc0 is closed first (tools/host_state_probe.py's order)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, fpng_amd
mode = sys.argv[1]
w, h, c = 7680, 4320, 4
img = fpng_amd.synth_image("grad", w, h, c)
out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
def measure(enc):
    s = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); enc.encode_host_into(img, w, h, c, out, 0); s = min(s, time.perf_counter() - t0)
    return s * 1e3
e0 = fpng_amd.Encoder(device=0, stream="own")
a = measure(e0)
if mode == "close":
    e0.close()
e1 = fpng_amd.Encoder(device=0, stream="own")
b = measure(e1)
c0 = measure(e0) if mode == "keep" else float("nan")
print(f"{mode}: enc0 {a:.2f} | enc1 {b:.2f} | enc0 again {c0:.2f}")
