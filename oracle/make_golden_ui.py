"""Golden vectors for screenshot-like content (TEST INFRASTRUCTURE): the images of tests/ui_images.py (glyph rows, flat panels with
anti-aliased edges, ordered-dither gradients, alpha mattes with long exact runs) through the UNMODIFIED reference encoder
(oracle/_ref/libfpng_ref.so), flags 0 and 1.

Run in the dev container (where /root/reference exists):   python oracle/make_golden_ui.py
Writes tests/golden/ui.json: per image and flags: size, sha256 of the reference's file, sha256 of the pixels."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from cpu_ref import ref
    import ui_images
    r = ref()
    out = {}
    for name, (img, w, h, c) in ui_images.all_images().items():
        e = {"w": w, "h": h, "c": c, "pixels_sha256": hashlib.sha256(img.tobytes()).hexdigest(), "flags": {}}
        for flags in (0, 1):
            png = r.encode(img, w, h, c, flags)
            st, px, *_ = r.decode(png, c)
            assert st == 0 and px.tobytes() == img.tobytes()
            e["flags"][str(flags)] = {"size": len(png), "sha256": hashlib.sha256(png).hexdigest()}
            print(name, flags, len(png), f"{len(png) / img.size:.4f} of raw")
        out[name] = e
    with open(os.path.join(ROOT, "tests", "golden", "ui.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
