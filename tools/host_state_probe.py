"""Is the host path's fast/slow state a property of the process or of the encoder's copy streams?  Three encoders one after the
other in one process, each measured (single streamed call, host_batch); run in several fresh processes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, fpng_amd
w, h, c = 7680, 4320, 4
imgs = [fpng_amd.synth_image("grad", w, h, c, seed=12345 + i) for i in range(6)]
outs = [np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8) for _ in range(6)]
line = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    enc = fpng_amd.Encoder(device=0, stream="own")
    s = b = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); enc.encode_host_into(imgs[0], w, h, c, outs[0], 0); s = min(s, time.perf_counter() - t0)
        t0 = time.perf_counter(); enc.encode_host_batch(imgs, 0, outs=outs); b = min(b, (time.perf_counter() - t0) / 6)
    line.append(f"enc{k}: single {s*1e3:.2f} batch {b*1e3:.2f}")
    enc.close()
print(" | ".join(line))
