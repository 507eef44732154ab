import os, sys, time, mmap
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, fpng_amd
w, h, c = 7680, 4320, 4
mode = sys.argv[1]
def buf(n, huge):
    m = mmap.mmap(-1, (n + 4095) & ~4095)
    m.madvise(mmap.MADV_HUGEPAGE if huge else mmap.MADV_NOHUGEPAGE)
    a = np.frombuffer(m, dtype=np.uint8)[:n]
    return a, m
img0 = fpng_amd.synth_image("grad", w, h, c).reshape(-1)
img, m1 = buf(img0.size, mode[0] == "H"); img[:] = img0
out, m2 = buf(fpng_amd.max_encoded_size(w, h, c), mode[1] == "H"); out[:] = 0
enc = fpng_amd.Encoder(device=0, stream="own")
ts = []
for _ in range(7):
    t0 = time.perf_counter(); n = enc.encode_host_into(img.reshape(h, w, c), w, h, c, out, 0); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
print("pixels", "huge" if mode[0] == "H" else "4K", "pages, output", "huge" if mode[1] == "H" else "4K", "pages:", ts, enc.last_host_bands())
