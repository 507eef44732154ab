"""Helper of tests/test_gpu_parity.py::test_lanes_follow_the_hardware_queues (run as a script, one process per case):
seven submissions in flight at once (1- and 2-pass alternating, three frames each, every submission its own output buffers) --
so every lane the library uses has a chain running next to the others' -- then every file against the checker's.
    python tests/lanes_check.py early|late      (late: the process has made HIP calls before the library is loaded)
prints "OK <hardware queues the library assumes> <lanes> <why: fpng_amd_runtime_info()'s source>"."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

if sys.argv[1] == "late":
    torch.cuda.init()
    torch.zeros(4, device="cuda").sum().item()
import fpng_amd
from cpu_ref import oracle

enc = fpng_amd.Encoder(device=0)
info = fpng_amd.runtime_info()
assert enc.lanes == info["lanes"]
specs = [("grad", 1920, 1080, 4), ("blocks", 1280, 720, 3), ("noise", 640, 480, 4)]
imgs = [fpng_amd.synth_image(k, w, h, c, seed=900 + i) for i, (k, w, h, c) in enumerate(specs)]
dev = [torch.from_numpy(im).cuda() for im in imgs]
expected = {fl: [oracle().encode(im, w, h, c, fl) for im, (k, w, h, c) in zip(imgs, specs)] for fl in (0, 1)}
subs = []
for i in range(7):
    outs = [torch.zeros(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device="cuda") for (k, w, h, c) in specs]
    enc.submit(dev, outs, i & 1)
    subs.append((enc.last_ticket, outs, i & 1))
for ticket, outs, fl in subs:
    for out, (size, mode, status), exp in zip(outs, enc.wait(ticket, len(specs)), expected[fl]):
        assert status == 0
        assert bytes(out[:size].cpu().numpy()) == exp, f"submission {ticket} flags {fl}: file differs from the checker's"
enc.close()
print("OK", info["hw_queues"], info["lanes"], info["hw_queue_source"], flush=True)
