import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, fpng_amd
from cpu_ref import oracle
enc = fpng_amd.Encoder(device=0)
def run(img, fl=0, name=""):
    h, w, c = img.shape
    (png,), (mode,) = enc.encode_tensors([torch.from_numpy(np.ascontiguousarray(img)).cuda()], fl)
    exp = oracle().encode(img, w, h, c, fl)
    if png == exp:
        print(name, w, h, c, "OK"); return
    a = np.frombuffer(png, np.uint8); b = np.frombuffer(exp, np.uint8)
    n = min(len(a), len(b)); d = np.nonzero(a[:n] != b[:n])[0]
    print(name, w, h, c, "DIFF sizes", len(a), len(b), "ndiff", len(d), "first", d[:12], "mode", mode)
    for i in d[:6]: print("   byte", i, "got %02x exp %02x" % (a[i], b[i]))
for args in [("solid",1,1,3),("solid",1,1,4),("solid",2,1,4),("solid",64,1,4),("solid",65,1,4),("solid",128,2,4),("solid",86,1,3),("solid",87,1,3),("grad",512,512,3)]:
    run(fpng_amd.synth_image(*args), 0, args[0])
