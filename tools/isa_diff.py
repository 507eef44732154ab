#!/usr/bin/env python3
"""Are a kernel file's compiled kernels the same instruction for instruction before and after a source change?

    python tools/isa_diff.py dump kernels.hip /tmp/before.json [-DNAME=VALUE ...]
    python tools/isa_diff.py compare /tmp/before.json /tmp/after.json

`dump` compiles the file device-only for gfx950 and stores, per kernel, the sha256 of its listing with the function numbers in
block labels (.LBB<n>_<k>) and debug notes taken out; `compare` names every kernel that is new, gone or different.  Used when code
is REMOVED from a frozen hot file: the kernels that stay must not move (no GPU needed)."""
import hashlib
import json
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from isa_loops import listing  # noqa: E402


def kernels(txt):
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*(?:;.*)?$", txt, re.M):
        name = m.group(1)
        end = txt.find(".Lfunc_end", m.end())
        if end < 0 or "s_endpgm" not in txt[m.end():end]:
            continue
        body = []
        for ln in txt[m.end():end].splitlines():
            ln = ln.split(";")[0].rstrip()
            if not ln.strip() or ln.strip().startswith((".loc", ".file", ".cfi", ".p2align")):
                continue
            body.append(re.sub(r"\.LBB\d+_", ".LBB_", ln))
        out[name] = {"sha256": hashlib.sha256("\n".join(body).encode()).hexdigest(), "lines": len(body)}
    return out


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        k = kernels(listing(sys.argv[2], sys.argv[4:]))
        json.dump(k, open(sys.argv[3], "w"), indent=1, sort_keys=True)
        print(f"{len(k)} kernels")
    else:
        a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
        same = [n for n in a if n in b and a[n] == b[n]]
        for n in sorted(set(a) | set(b)):
            if n not in b:
                print("gone      ", n)
            elif n not in a:
                print("new       ", n)
            elif a[n] != b[n]:
                print("DIFFERENT ", n, a[n]["lines"], "->", b[n]["lines"])
        print(f"{len(same)} kernels identical")
