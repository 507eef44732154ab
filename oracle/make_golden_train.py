"""Golden vectors of the TABLE TRAINING (TEST INFRASTRUCTURE): the reference built with -DFPNG_TRAIN_HUFFMAN_TABLES=1
(oracle/_ref/libfpng_ref_train.so: src/fpng_test.cpp:766-973 training_mode restated for images in memory + the reference's own
create_dynamic_block_prefix, src/fpng.cpp:910-988) on two small corpora.

Run in the dev container:   python oracle/make_golden_train.py   ->  tests/golden/train.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def corpora():
    """name -> (num_chans, [uint8 (h, w, c) images]); tests/test_train.py builds the same lists."""
    import fpng_amd
    import real_image
    from cpu_ref import ref
    v = real_image.variants(real_image.rgb_pixels(ref().decode if os.path.exists("/root/reference") else __import__("dropin").decode))
    rgb, ga = v["rgb"], v["rgba_ga"]
    opaque = [rgb, np.ascontiguousarray(rgb[100:400, 50:561]), fpng_amd.synth_image("grad", 512, 512, 3), fpng_amd.synth_image("blocks", 640, 200, 3),
              fpng_amd.synth_image("noise", 97, 33, 3), fpng_amd.synth_image("solid", 300, 20, 3), np.ascontiguousarray(rgb[::3, ::3])]
    alpha = [ga, np.ascontiguousarray(ga[::2, ::2]), fpng_amd.synth_image("grad", 800, 450, 4), fpng_amd.synth_image("blocks", 320, 240, 4),
             fpng_amd.synth_image("noise", 64, 48, 4), fpng_amd.synth_image("solid", 5, 3, 4)]
    return {"opaque": (3, opaque), "alpha": (4, alpha), "one_tiny_image": (4, [fpng_amd.synth_image("solid", 2, 1, 4)])}


def ref_train(imgs, c):
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfpng_ref_train.so"))
    L.ref_init()
    n = len(imgs)
    ptrs = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
    ws = (C.c_uint32 * n)(*[i.shape[1] for i in imgs])
    hs = (C.c_uint32 * n)(*[i.shape[0] for i in imgs])
    prefix = (C.c_uint8 * 4096)()
    plen, bb, bbs = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    codes = (C.c_uint32 * 288)()
    sizes = (C.c_uint8 * 288)()
    ok = L.ref_train(ptrs, ws, hs, n, c, prefix, 4096, C.byref(plen), C.byref(bb), C.byref(bbs), codes, sizes)
    assert ok
    return {"prefix": bytes(prefix[: plen.value]).hex(), "bit_buf": bb.value, "bit_buf_size": bbs.value, "codes": list(codes), "code_sizes": list(sizes)}


def main():
    out = {}
    for name, (c, imgs) in corpora().items():
        out[name] = dict(num_chans=c, n=len(imgs), **ref_train([np.ascontiguousarray(i) for i in imgs], c))
        print(name, c, len(imgs), "prefix", len(out[name]["prefix"]) // 2, "bytes +", out[name]["bit_buf_size"], "bits")
    with open(os.path.join(ROOT, "tests", "golden", "train.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
