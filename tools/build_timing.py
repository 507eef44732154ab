import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import torch, fpng_amd
enc = fpng_amd.Encoder(device=0, stream="own")
enc.set_profiling(True)
for (k, w, h, c) in [("grad", 3840, 2160, 4), ("blocks", 1920, 1080, 3)]:
    img = torch.from_numpy(fpng_amd.synth_image(k, w, h, c)).cuda()
    out = torch.empty(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        enc.submit([img], [out], 1); enc.finish(1)
    buf = (C.c_uint32 * 8)()
    enc.lib.fpng_amd_debug_peek(enc.h, 0, buf, 8)
    print(k, "cycles table288/seq/rle/table19/header/publish:", list(buf)[:6], "total us ~", round(sum(list(buf)[:6]) / 2100, 1))
    buf[7] = 0xFEED
    enc.lib.fpng_amd_debug_peek(enc.h, 0, buf, 8)
    print("   inside table288: sort/merge/depths/kraft/codes:", list(buf)[:5])
