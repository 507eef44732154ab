import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, fpng_amd
mode = sys.argv[1]
w, h, c = 7680, 4320, 4
imgs = [fpng_amd.synth_image("grad", w, h, c, seed=12345 + i) for i in range(6)]
outs = [np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8) for _ in range(6)]
enc = fpng_amd.Encoder(device=0, stream="own")
def tb():
    b = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); enc.encode_host_batch(imgs, 0, outs=outs); b = min(b, (time.perf_counter() - t0) / 6)
    return b * 1e3
def ts():
    b = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); enc.encode_host_into(imgs[0], w, h, c, outs[0], 0); b = min(b, time.perf_counter() - t0)
    return b * 1e3
if mode.startswith("P"):
    pw, ph, pc = {"P1": (512, 512, 3), "P2": (1920, 1080, 3), "P3": (3840, 2160, 4), "P0": (64, 64, 3), "P4": (7680, 4320, 4), "P5": (7680, 4320, 4)}[mode]
    pim = fpng_amd.synth_image("grad", pw, ph, pc)
    pout = np.empty(fpng_amd.max_encoded_size(pw, ph, pc), dtype=np.uint8)
    enc.encode_host_into(pim, pw, ph, pc, pout, 0)
    if mode == "P5":  # ... and every frame / output buffer of the batch once through the serial path
        for im, o in zip(imgs, outs):
            enc.encode_host_into(im, w, h, c, o, 0)
    print(mode, f": primed with one {pw}x{ph}x{pc} call: host_batch", round(tb(), 3), "| single", round(ts(), 3), "| host_batch", round(tb(), 3))
elif mode == "A":
    print("A: host_batch first:", round(tb(), 3), "| single", round(ts(), 3), "| host_batch again", round(tb(), 3))
else:
    print(mode, ": single first:", round(ts(), 3), "| host_batch", round(tb(), 3), "| single", round(ts(), 3), "| host_batch", round(tb(), 3))
