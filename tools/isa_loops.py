#!/usr/bin/env python3
"""Static instruction counts per loop of a kernel, from a device-only assembly listing (no GPU needed):

    python tools/isa_loops.py decode.hip dec_emit_kernel [-DNAME=VALUE ...] [--blocks BBk_n]

Every basic block of the listing carries the compiler's note of the innermost loop it belongs to ("in Loop: Header=BBk_n
Depth=d"); the block's instructions are counted into that loop, by class: VALU (v_*), SALU (s_* without waits / branches / nops),
LDS (ds_*), VMEM (global_* / buffer_* / flat_* / scratch_*), branches, waits.  A kernel whose time is vector-issue bound -- a wave64
VALU instruction occupies a SIMD for four cycles -- runs as long as its hot loop's VALU count says: dec_emit_kernel's ~4 300 VALU
instructions per wave and subsequence x 4 cycles x the waves a SIMD gets are its measured 0.85 ms, so a change's
effect on such a kernel can be read here before a GPU is asked."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fpng_amd", "csrc")


def listing(src, defs):
    with tempfile.TemporaryDirectory() as t:
        out = os.path.join(t, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "--cuda-device-only", "-S", "-I", os.path.join(ROOT, "include"),
                               "-I", CSRC, "-o", out, "-x", "hip", os.path.join(CSRC, src)] + defs, stderr=subprocess.DEVNULL)
        return open(out).read()


def classify(mn):
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if mn.startswith("s_load") or mn.startswith("s_buffer_load"):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    return "other"


def kernel_text(txt, name):
    m = re.search(r"^(_Z\w*" + re.escape(name) + r"\w*):\s*(?:;.*)?$", txt, re.M)
    if not m:
        raise SystemExit(f"no kernel matching {name}")
    end = txt.index(".Lfunc_end", m.end())
    return m.group(1), txt[m.end():end]


def loops(body):
    """-> {header: {"depth": d, "parent": header or None, counts...}}, block order kept"""
    res = {}
    cur = None  # innermost loop of the current block (None = outside every loop)
    order = []
    for line in body.splitlines():
        s = line.strip()
        if not s:
            continue
        lab = re.match(r"^(\.LBB\d+_\d+):", s)
        note = None
        if lab or s.startswith("; %bb."):
            if "This" in s and "Loop Header" in s:  # the header block's own note follows on the next comment lines
                pass
            m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", s)
            if m:
                cur = m.group(1)
                res.setdefault(cur, {"depth": int(m.group(2))})
                if cur not in order:
                    order.append(cur)
            elif lab and "Loop" not in s:
                cur = None
            if lab and "Parent Loop" in s:  # a header: its name is the label itself
                cur = lab.group(1)[2:]
                res.setdefault(cur, {"depth": 0})
                if cur not in order:
                    order.append(cur)
            continue
        if s.startswith(";"):
            m = re.search(r"=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", s)
            if m and cur is not None:
                res[cur]["depth"] = int(m.group(1))
            continue
        if s.startswith("."):
            continue
        mn = s.split()[0]
        k = classify(mn)
        if cur is not None:
            res[cur][k] = res[cur].get(k, 0) + 1
    return res, order


def blocks_of(body, header):
    """Per basic block of loop `header` (child loops excluded), in listing order: label, VALU, SALU, LDS, VMEM -- the straight-line
    path of a hot loop is the blocks between its header and the first rarely taken branch."""
    cur, name, rows = None, None, []
    for line in body.splitlines():
        s = line.strip()
        lab = re.match(r"^(\.LBB\d+_\d+):", s)
        if lab or s.startswith("; %bb."):
            m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth", s)
            cur = m.group(1) if m else (lab.group(1)[2:] if lab and "Parent Loop" in s else None)
            name = lab.group(1) if lab else s.split()[1].rstrip(":")
            if cur == header:
                rows.append([name, {}])
            continue
        if cur != header or not s or s[0] in ";.":
            continue
        k = classify(s.split()[0])
        rows[-1][1][k] = rows[-1][1].get(k, 0) + 1
    return rows


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    src, name = sys.argv[1], sys.argv[2]
    defs = [a for a in sys.argv[3:] if a.startswith("-D")]
    full, body = kernel_text(listing(src, defs), name)
    if "--blocks" in sys.argv:
        h = sys.argv[sys.argv.index("--blocks") + 1]
        print(f"{full[:100]}  {' '.join(defs)}  loop {h}")
        for nm, c in blocks_of(body, h):
            print(f"{nm:14s} VALU {c.get('valu', 0):4d} SALU {c.get('salu', 0):4d} LDS {c.get('lds', 0):3d} VMEM {c.get('vmem', 0):3d} branch {c.get('branch', 0):2d}")
        return
    # a loop header that is the kernel's first labelled block of a loop carries "=>This Loop Header" on its own label line
    res, order = loops(body)
    print(f"{full[:100]}  {' '.join(defs)}")
    print(f"{'loop':12s} {'depth':>5s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'branch':>6s} {'wait':>5s}")
    tot = {}
    for h in order:
        r = res[h]
        print(f"{h:12s} {r.get('depth', 0):5d} {r.get('valu', 0):6d} {r.get('salu', 0):6d} {r.get('lds', 0):5d} {r.get('vmem', 0):5d} {r.get('branch', 0):6d} {r.get('wait', 0):5d}")
        for k, v in r.items():
            if k != "depth":
                tot[k] = tot.get(k, 0) + v
    print(f"{'(all loops)':12s} {'':5s} {tot.get('valu', 0):6d} {tot.get('salu', 0):6d} {tot.get('lds', 0):5d} {tot.get('vmem', 0):5d} {tot.get('branch', 0):6d} {tot.get('wait', 0):5d}")


if __name__ == "__main__":
    main()
