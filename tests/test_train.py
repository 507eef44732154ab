"""Table training (reference src/fpng_test.cpp:766-973 + src/fpng.cpp:909-988 under FPNG_TRAIN_HUFFMAN_TABLES):
the oracle's restatement (CPU) and the HIP implementation (-m gpu) against the reference built with
-DFPNG_TRAIN_HUFFMAN_TABLES=1 (tests/golden/train.json, made by oracle/make_golden_train.py)."""
import json
import os
import sys

import numpy as np
import pytest

from cpu_ref import ROOT, oracle

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _gold():
    with open(os.path.join(ROOT, "tests", "golden", "train.json")) as f:
        return json.load(f)


def _corpora():
    import make_golden_train
    return make_golden_train.corpora()


def _same(got, want, what):
    for k in ("prefix", "bit_buf", "bit_buf_size", "code_sizes", "codes"):
        assert got[k] == want[k], (what, k)


def test_oracle_training_vs_reference(built_lib):
    g = _gold()
    for name, (c, imgs) in _corpora().items():
        _same(oracle().train_tables(imgs, c), g[name], name)


@pytest.mark.gpu
def test_hip_training_vs_reference(built_lib):
    import torch
    import fpng_amd
    g = _gold()
    enc = fpng_amd.Encoder(device=0)
    for name, (c, imgs) in _corpora().items():
        got = enc.train_tables([torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs])
        _same(got, g[name], name)
    enc.close()


@pytest.mark.gpu
def test_trained_table_is_a_valid_one_pass_table(built_lib):
    """What training is for: the prefix + codes describe a complete Deflate dynamic block header in which every literal,
    the end-of-block symbol and every run length the coder can emit has a code (zlib accepts a stream built from it)."""
    import zlib
    import torch
    import fpng_amd
    enc = fpng_amd.Encoder(device=0)
    c, imgs = _corpora()["opaque"]
    t = enc.train_tables([torch.from_numpy(np.ascontiguousarray(i)).cuda() for i in imgs])
    sizes, codes = t["code_sizes"], t["codes"]
    assert all(sizes[i] > 0 for i in range(257)) and max(sizes) <= 12
    # a stream: prefix, pending bits, the literals 'f','p','n','g', end of block, Adler-32
    bits, nbits = t["bit_buf"], t["bit_buf_size"]
    payload = b"fpng"
    for sym in list(payload) + [256]:
        bits |= codes[sym] << nbits
        nbits += sizes[sym]
    body = bytes.fromhex(t["prefix"]) + bits.to_bytes((nbits + 7) // 8, "little") + zlib.adler32(payload).to_bytes(4, "big")
    assert zlib.decompress(body) == payload
    enc.close()


@pytest.mark.gpu
def test_command_line_training_mode(built_lib, tmp_path):
    """fpng_amd_test -t @list (the reference's `fpng_test -t`, fpng_test.cpp:766-973): opaque and translucent files trained
    separately, printed in the reference's form; the numbers are those of the library call on the same pixels."""
    import re
    import subprocess
    import torch
    import fpng_amd
    corpus = str(tmp_path / "corpus")
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_corpus.py"), corpus], env=dict(env, FPNG_CORPUS_DROPIN="1"))
    names = sorted(os.path.join(corpus, f) for f in os.listdir(corpus) if not f.startswith("ui_"))  # (the photograph's files: 8 opaque, 1 translucent)
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write("\n".join(names) + "\n")
    r = subprocess.run([os.path.join(ROOT, "fpng_amd", "lib", "fpng_amd_test"), "-t", "@" + lst], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-500:]
    assert "Total alpha files: 1" in r.stdout and "Total opaque files: 8" in r.stdout
    m = re.search(r"g_dyn_huff_3\[\] = \{([^}]*)\};\s*const uint32_t DYN_HUFF_3_BITBUF = (\d+), DYN_HUFF_3_BITBUF_SIZE = (\d+)", r.stdout)
    assert m and "g_dyn_huff_4_codes[288]" in r.stdout
    prefix = bytes(int(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip())
    assert prefix[:2] == b"\x78\x01" and int(m.group(3)) < 8
