"""Random submissions kept in flight on every lane (round 5): batches of 1-6 frames of random shapes, contents and modes (1-pass, 2-pass,
stored), up to eight submissions in flight, every file compared with the checker's once its submission is done.
    python tools/gpu_lanes_stress.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fpng_amd
import numpy as np
import torch
from cpu_ref import oracle

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
enc = fpng_amd.Encoder(device=0)  # ordered behind torch's stream: the zero fill of the output buffers comes first
kinds = ["grad", "blocks", "noise", "solid"]
# a pool of frames with their expected files (the checker is the slow part)
pool = []
t_pool = time.time()
while len(pool) < 60 and time.time() - t_pool < 40:
    c = int(rng.choice([3, 4]))
    w = int(rng.choice([1, 7, 64, 255, 256, 257, 640, 1000, 1920, 2048, 3840, 4099]))
    h = int(rng.choice([1, 3, 16, 100, 480, 1080])) if w < 3000 else int(rng.choice([1, 5, 64, 200]))
    kind = kinds[int(rng.integers(len(kinds)))]
    im = fpng_amd.synth_image(kind, w, h, c, seed=int(rng.integers(1 << 30)))
    exp = {fl: oracle().encode(im, w, h, c, fl) for fl in (0, 1, 2)}
    pool.append((torch.from_numpy(im).cuda(), w, h, c, kind, exp))
print(f"{len(pool)} frames in the pool", flush=True)
flying = []  # (ticket, frames, outs, flags)
n_sub = n_files = 0
t0 = time.time()


def check(ticket, frames, outs, fl):
    global n_files
    for (t, w, h, c, kind, exp), out, (size, mode, status) in zip(frames, outs, enc.wait(ticket, len(frames))):
        assert status == 0, f"status {status}"
        got = bytes(out[:size].cpu().numpy())
        if got != exp[fl]:
            a, b = np.frombuffer(got, np.uint8), np.frombuffer(exp[fl], np.uint8)
            n = min(len(a), len(b))
            d = np.nonzero(a[:n] != b[:n])[0]
            others = [k for k in (0, 1, 2) if got == exp[k]]
            raise SystemExit(f"MISMATCH: ticket {ticket}, {kind} {w}x{h}x{c} flags {fl} mode {mode}: {len(got)} vs {len(exp[fl])} bytes, {len(d)} bytes differ, first at {d[:8].tolist()}, "
                             f"last at {d[-4:].tolist()}; got {got[d[0]:d[0] + 8].hex() if len(d) else ''} expected {exp[fl][d[0]:d[0] + 8].hex() if len(d) else ''}; equals the file of flags {others}; "
                             f"batch: {[(k2, w2, h2, c2) for (_, w2, h2, c2, k2, _) in frames]}")
        n_files += 1


while time.time() - t0 < secs:
    frames = [pool[int(rng.integers(len(pool)))] for _ in range(int(rng.integers(1, 7)))]
    fl = int(rng.choice([int(v) for v in os.environ.get("STRESS_FLAGS", "0,0,1,1,2").split(",")]))
    outs = [torch.zeros(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device="cuda") for (_, w, h, c, _, _) in frames]
    enc.submit([f[0] for f in frames], outs, fl)
    flying.append((enc.last_ticket, frames, outs, fl))
    n_sub += 1
    while len(flying) >= int(rng.integers(1, 9)):  # drain down to a random depth
        check(*flying.pop(0))
for f in flying:
    check(*f)
enc.close()
print(f"lanes stress: {n_sub} submissions, {n_files} files, no mismatch, {time.time() - t0:.0f} s", flush=True)
