"""Debug helper for the fused pipeline: which images of a batch differ from the golden vectors, and how the
kernel time scales with the number of persistent blocks (FPNG_AMD_FUSED_BLOCKS)."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpng_amd

what = sys.argv[1]
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "batches.json")))
enc = fpng_amd.Encoder(device=0)
if what == "check":
    name, flags, n = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    s = gold[name]
    w, h, c = s["w"], s["h"], s["c"]
    imgs = [torch.from_numpy(fpng_amd.synth_image(s["kind"], w, h, c, seed=s["seed0"] + i)).cuda() for i in range(n)]
    outs = [torch.empty(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device="cuda") for _ in range(n)]
    for rep in range(3):
        enc.submit(imgs, outs, flags)
        res = enc.finish(n)
        bad = []
        for i, (size, mode, status) in enumerate(res):
            ok = status == 0 and size == s["flags"][str(flags)]["sizes"][i] and \
                hashlib.sha256(outs[i][:size].cpu().numpy().tobytes()).hexdigest() == s["flags"][str(flags)]["sha256"][i]
            if not ok:
                bad.append((i, size, mode, status))
        print(name, "rep", rep, "bad:", bad[:20], len(bad), flush=True)
elif what == "time":
    w, h, c, n = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    imgs = [torch.from_numpy(fpng_amd.synth_image("grad", w, h, c, seed=12345 + i)).cuda() for i in range(n)]
    outs = [torch.empty(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device="cuda") for _ in range(n)]
    enc.set_profiling(True)
    for rep in range(4):
        enc.submit(imgs, outs, 0)
        enc.finish(n)
        print(os.environ.get("FPNG_AMD_FUSED_BLOCKS"), dict(zip(enc.phase_names(), [round(x, 4) for x in enc.last_phase_ms()])), flush=True)
        import ctypes as C
        buf = (C.c_uint32 * 512)()
        enc.lib.fpng_amd_debug_peek(enc.h, 0, buf, 512)
        tick = [buf[256 + 32 * k] for k in range(8)]
        note = list(buf[481:494])
        if "timing" in os.environ.get("FPNG_AMD_LIB", ""):
            tb = (C.c_uint32 * 16)()
            enc.lib.fpng_amd_debug_peek(enc.h, 0, tb, 16)
            cyc = [tb[2 * k] | (tb[2 * k + 1] << 32) for k in range(6)]
            tot = sum(cyc) or 1
            print("  cycles ticket/walk/wait/lookback/done/copy %:", [round(100 * v / tot, 1) for v in cyc], "total Mcycles", round(tot / 1e6, 2), flush=True)
        print("  tickets", tick, "abort", buf[511], "note", [hex(v) for v in note] if note[0] == 0xDEB06 else None, flush=True)
