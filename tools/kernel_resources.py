#!/usr/bin/env python3
"""Register / LDS / spill counts of every kernel (the metadata at the end of a device-only assembly listing; no GPU needed):
    python tools/kernel_resources.py [-DNAME=VALUE ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fpng_amd", "csrc")
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
with tempfile.TemporaryDirectory() as t:
    for src in ("kernels.hip", "decode.hip"):
        co = os.path.join(t, src + ".s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"),
                               "-I", CSRC, "-o", co, "-x", "hip", os.path.join(CSRC, src)] + defs, stderr=subprocess.DEVNULL)
        txt = open(co).read()
        txt = txt[txt.index(".amdgpu_metadata"):]
        for blk in txt.split("- .agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
            name = re.sub(r"^_ZN8fpng_amd12_GLOBAL__N_1\d+", "", g("name"))[:58]
            print(f"{name:60s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s} "
                  f"vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s}")
