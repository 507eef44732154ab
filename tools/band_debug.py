"""Debug helper: per band, compare the GPU's window with the oracle stand-in's (tests/band_backend.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fpng_amd
from fpng_amd import sharded
from cpu_ref import fuzz_image, oracle
from band_backend import OracleBandBackend

enc = fpng_amd.Encoder(device=0)
gb = sharded.GpuBandBackend(enc)
rng = np.random.default_rng(21)
bad = 0
for case in range(400):
    img, w, h, c = fuzz_image(rng, force_dims=(int(rng.integers(1, 200)), int(rng.integers(2, 30))))
    nb = int(rng.integers(2, 5))
    cuts = sorted(set([0, h] + [int(v) for v in rng.integers(0, h + 1, nb - 1)]))
    fl = case & 1
    t = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    ob = OracleBandBackend(img)
    bands = list(zip(cuts[:-1], cuts[1:]))
    above = lambda y0: t[y0 - 1] if y0 else None
    hist = None
    if fl:
        hist = sum(gb.hist(t[a:b], above(a), w, c, a, b, h).to(torch.int64) for a, b in bands).to(torch.int32)
        oh = sum(ob.hist(None, None, w, c, a, b, h).to(torch.int64) for a, b in bands).to(torch.int32)
        if not torch.equal(hist.cpu(), oh):
            print("HIST differs", case, w, h, c); bad += 1; continue
    gstats = [gb.encode(t[a:b], above(a), w, c, a, b, h, fl, hist) for a, b in bands]
    ostats = [ob.encode(None, None, w, c, a, b, h, fl, hist.cpu() if hist is not None else None) for a, b in bands]
    if [tuple(vars(s).values()) for s in gstats] != [tuple(vars(s).values()) for s in ostats]:
        print("STATS differ", case, (w, h, c), cuts, fl, gstats, ostats); bad += 1; continue
    plan = sharded.plan_bands(gstats, w, h, c, gstats[0].first_token_bit, gstats[0].eob_bits, not fl)
    if plan.stored:
        continue
    for i, (a, b) in enumerate(bands):
        gb.encode(t[a:b], above(a), w, c, a, b, h, fl, hist)
        goff, gwin = gb.place(plan.start_bits[i], plan.zlib_size, gstats[i].token_bits, t.device)
        ob.encode(None, None, w, c, a, b, h, fl, hist.cpu() if hist is not None else None)
        try:
            ooff, owin = ob.place(plan.start_bits[i], plan.zlib_size, ostats[i].token_bits, "cpu")
        except AssertionError:
            print("ASSERT", case, (w, h, c), cuts, fl, i, plan.start_bits, ob._band["ftb"], ob._band["first"], gstats, ostats); bad += 1; break
        g = gwin.cpu().numpy().copy(); o = owin.numpy().copy()
        if i == 0:
            g[:58] = 0; o[:58] = 0
        if goff != ooff or g.size != o.size or (g != o).any():
            d = int(np.argmax(g[:min(g.size, o.size)] != o[:min(g.size, o.size)])) if g.size and o.size else -1
            print("WINDOW differs", case, (w, h, c), cuts, "flags", fl, "band", i, "off", goff, ooff, "size", g.size, o.size, "first diff", d,
                  g[d:d + 4].tobytes().hex(), o[d:d + 4].tobytes().hex(), "start_bit", plan.start_bits[i], "bits", gstats[i].token_bits)
            bad += 1
            break
print("done, bad =", bad)
