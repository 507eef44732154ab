#!/usr/bin/env python
"""A long differential campaign of the GPU decoder's LOGIC on the CPU (no GPU needed): tests/cpp/decode_emul.cpp runs the kernels' own
per-thread code (fpng_amd/csrc/decode_core.h) thread by thread; the judge is the reference's decoder (oracle/_ref), status AND pixels.

    python tools/emul_campaign.py <seconds> [seed]           # one process; run several with different seeds
    python tools/emul_campaign.py --large <seconds> [seed]   # megapixel images, one local token edit per file

Content: the fuzz generator's images, crops of the screenshot-like generators, periodic stripes / tiles, crops of the photograph,
flat images; 1-pass and 2-pass files; random workgroup size / lead-in / tile parameters (small ones put many borders and seams into
small images).  Every valid file must decode to its pixels; then damaged copies (bit flips stratified over the stream, truncations,
header edits, token-bit flips, spliced streams) must get the reference's status and, where it decodes them, its pixels; so must
files whose TOKENS were edited (tests/token_mutator.py: matches lengthened, split, moved to a row's first pixel or off a pixel
boundary, filter literals changed, the end-of-block symbol moved ...), by the kernels' logic and by the drop-in's CPU decoder."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_decode_model as M  # noqa: E402
import token_mutator as TM  # noqa: E402
import container_mutator as CM  # noqa: E402
import header_mutator as HM  # noqa: E402
import ctypes as C  # noqa: E402
import ui_images  # noqa: E402
from cpu_ref import fuzz_image, oracle, ref  # noqa: E402


def content(rng, photo):
    k = int(rng.integers(0, 7))
    if k == 0:
        return fuzz_image(rng)
    if k == 1:
        return fuzz_image(rng, force_dims=(int(rng.integers(100, 900)), int(rng.integers(2, 24))))
    if k == 2:  # a crop of a screenshot-like image
        c = int(rng.choice([3, 4]))
        gen = [ui_images.glyphs, ui_images.panels, ui_images.dither][int(rng.integers(0, 3))]
        w, h = int(rng.integers(64, 700)), int(rng.integers(16, 90))
        a = gen(w, h, c, seed=int(rng.integers(1, 1 << 30)))
        return np.ascontiguousarray(a).reshape(-1), w, h, c
    if k == 3:  # periodic stripes / tiles, sometimes sprinkled
        period, c, w, h = int(rng.integers(1, 40)), int(rng.choice([3, 4])), int(rng.integers(200, 3000)), int(rng.integers(8, 80))
        pal = rng.integers(0, 256, (period, c), dtype=np.uint8)
        img = (pal[np.arange(w) % period][None] + (np.arange(h)[:, None, None] * int(rng.integers(0, 9))).astype(np.uint8)).astype(np.uint8)
        if rng.random() < 0.3:
            img.reshape(-1, c)[rng.integers(0, w * h, 10)] = rng.integers(0, 256, (10, c), dtype=np.uint8)
        return np.ascontiguousarray(img).reshape(-1), w, h, c
    if k == 4 and photo is not None:  # a crop of the photograph
        H, W, _ = photo.shape
        w, h = int(rng.integers(32, 500)), int(rng.integers(8, 60))
        x0, y0 = int(rng.integers(0, W - w)), int(rng.integers(0, H - h))
        c = int(rng.choice([3, 4]))
        a = photo[y0:y0 + h, x0:x0 + w]
        if c == 4:
            a = np.concatenate([a, np.full((h, w, 1), int(rng.integers(0, 256)), np.uint8)], axis=2)
        return np.ascontiguousarray(a).reshape(-1), w, h, c
    if k == 5:  # flat / two-colour: long runs, the wave-filled kind
        c, w, h = int(rng.choice([3, 4])), int(rng.integers(100, 5000)), int(rng.integers(2, 40))
        img = np.empty((h, w, c), np.uint8)
        img[:] = rng.integers(0, 256, c, dtype=np.uint8)
        if rng.random() < 0.5:
            x = int(rng.integers(0, w))
            img[:, x:] = rng.integers(0, 256, c, dtype=np.uint8)
        return img.reshape(-1), w, h, c
    import fpng_amd
    kind = ["grad", "blocks", "noise", "solid"][int(rng.integers(0, 4))]
    c, w, h = int(rng.choice([3, 4])), int(rng.integers(8, 1200)), int(rng.integers(1, 50))
    return np.asarray(fpng_amd.synth_image(kind, w, h, c, seed=int(rng.integers(0, 1 << 30)))).reshape(-1), w, h, c


def damage(rng, png, other):
    bad = bytearray(png)
    kind = int(rng.integers(0, 8))
    n = len(bad)
    if kind == 0:
        i = int(rng.integers(0, n)); bad[i] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:
        bad = bad[: int(rng.integers(1, n))]
    elif kind == 2:
        i = int(rng.integers(58, min(140, n))); bad[i] = int(rng.integers(0, 256))
    elif kind == 3:
        i = int(rng.integers(58, n)); bad[i] = int(rng.integers(0, 256))
    elif kind == 4:
        i = int(rng.integers(min(125, n - 1), n)); bad[i] ^= 1 << int(rng.integers(0, 8))
    elif kind == 5:  # several flips spread over the token bits
        for _ in range(int(rng.integers(2, 6))):
            i = int(rng.integers(min(125, n - 1), n)); bad[i] ^= 1 << int(rng.integers(0, 8))
    elif kind == 6:  # a stretch of the stream zeroed / set
        i = int(rng.integers(58, n)); j = min(n - 16, i + int(rng.integers(1, 40)))
        for q in range(i, max(i, j)):
            bad[q] = 0 if rng.random() < 0.5 else 255
    else:  # the tail of another file's stream spliced in (same container head)
        if other is not None and len(other) > 200 and n > 200:
            i = int(rng.integers(130, min(n, len(other)) - 20))
            bad[i:n - 16] = other[i:i + (n - 16 - i)].ljust(n - 16 - i, b"\0")[: n - 16 - i]
    return kind, bytes(bad)


def large_images(rng, photo):
    """megapixel content: streams that span many workgroups of the kernels' own size"""
    import fpng_amd
    while True:
        k = int(rng.integers(0, 5))
        c = int(rng.choice([3, 4]))
        if k == 0:
            w, h = int(rng.integers(600, 2000)), int(rng.integers(300, 1100))
            yield np.asarray(fpng_amd.synth_image("grad", w, h, c, seed=int(rng.integers(0, 1 << 30)))).reshape(-1), w, h, c
        elif k == 1 and photo is not None:
            t = np.tile(photo, (2, 2, 1))
            w, h = int(rng.integers(500, t.shape[1])), int(rng.integers(300, t.shape[0]))
            yield np.ascontiguousarray(t[:h, :w]).reshape(-1), w, h, 3
        elif k == 2:
            gen = [ui_images.glyphs, ui_images.panels, ui_images.dither][int(rng.integers(0, 3))]
            w, h = int(rng.integers(800, 1920)), int(rng.integers(400, 1080))
            yield np.ascontiguousarray(gen(w, h, c, seed=int(rng.integers(1, 1 << 30)))).reshape(-1), w, h, c
        elif k == 3:
            w, h = int(rng.integers(800, 1920)), int(rng.integers(400, 1080))
            yield np.ascontiguousarray(ui_images.matte(w, h, seed=int(rng.integers(1, 1 << 30)))).reshape(-1), w, h, 4
        else:
            w, h = int(rng.integers(600, 2000)), int(rng.integers(300, 1100))
            yield np.asarray(fpng_amd.synth_image("blocks", w, h, c, seed=int(rng.integers(0, 1 << 30)))).reshape(-1), w, h, c


def main_large(secs, seed):
    """--large: ONE local token edit per file of a megapixel image (tests/token_mutator.py: LargeStream, mutate_large), the kernels'
    own workgroup size"""
    rng = np.random.default_rng(seed)
    judge = ref().decode
    import dropin
    import real_image
    os.environ["FPNG_AMD_DECODE_CPU"] = "1"
    photo = real_image.rgb_pixels(judge)
    t_end = time.time() + secs
    stats, n, images = {}, 0, 0
    for img, w, h, c in large_images(rng, photo):
        if time.time() > t_end:
            break
        flags = int(rng.integers(0, 2))
        png = oracle().encode(img, w, h, c, flags)
        if M.plan(png)[1]:
            continue
        images += 1
        s = TM.LargeStream(png, M.plan, M.emul())
        for _ in range(6):
            name, f = TM.mutate_large(s, rng)
            if f is None:
                continue
            desired = int(rng.choice([3, 4]))
            st_r, out_r, *_ = judge(f, desired)
            st_c, out_c, *_ = dropin.decode(f, desired)
            try:
                st_m, out_m, *_ = M.emul_decode(f, desired, M.CONFIGS[0])
            except AssertionError as ex:
                print(f"EMULATOR ERROR {ex}: seed {seed} {w}x{h}x{c} flags {flags} edit {name}", flush=True)
                open(f"/tmp/emul_campaign_err_{seed}_L{n}.png", "wb").write(f)
                continue
            und = st_m == M.UNDECIDED
            if und:
                st_m, out_m = st_c, out_c
            key = (name, st_r == 0, und)
            stats[key] = stats.get(key, 0) + 1
            n += 1
            if not all(st == st_r and (st_r != 0 or np.array_equal(np.asarray(out_r)[: o.size], o)) for st, o in ((st_m, out_m), (st_c, out_c))):
                print(f"MISMATCH large file: seed {seed} {w}x{h}x{c} flags {flags} edit {name} desired {desired} reference {st_r} emulator {st_m} cpu tier {st_c}", flush=True)
                open(f"/tmp/emul_campaign_fail_{seed}_L{n}.png", "wb").write(f)
    print(f"seed {seed} (large): {images} images of 0.2-2 MP, {n} files with one edited token each; (edit, accepted by the reference, left to the CPU decoder): count = {dict(sorted(stats.items()))}", flush=True)


def main():
    if "--large" in sys.argv:
        a = [v for v in sys.argv[1:] if v != "--large"]
        return main_large(float(a[0]) if a else 60.0, int(a[1]) if len(a) > 1 else 1)
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    judge = ref().decode
    import dropin
    os.environ["FPNG_AMD_DECODE_CPU"] = "1"
    cpu_tier = dropin.decode
    photo = None
    try:
        import real_image
        photo = real_image.rgb_pixels(judge)
    except Exception:
        pass
    t_end = time.time() + secs
    files = valid = damaged = rejected = undecided = edits = edits_accepted = edits_undecided = containers = containers_accepted = tables = tables_accepted = 0
    by_kind = {}
    prev_png = None
    while time.time() < t_end:
        img, w, h, c = content(rng, photo)
        flags = int(rng.integers(0, 2))
        png = oracle().encode(img, w, h, c, flags)
        files += 1
        cfgs = [M.CONFIGS[0], M.CONFIGS[int(rng.integers(1, len(M.CONFIGS)))],
                (int(rng.choice([2, 3, 4, 8, 16, 64, 512])), int(rng.choice([0, 32, 64, 128])), int(rng.choice([4, 52, 64, 100, 1024, 18432])))]
        for cfg in cfgs:
            for desired in (3, 4):
                st, px, ww, hh, cc, stats = M.emul_decode(png, desired, cfg)
                if st != 0 or not np.array_equal(px, M.expected_pixels(img, w, h, c, desired)):
                    print(f"MISMATCH valid file: seed {seed} file {files} {w}x{h}x{c} flags {flags} cfg {cfg} desired {desired} status {st}", flush=True)
                    open(f"/tmp/emul_campaign_fail_{seed}_{files}.png", "wb").write(png)
                valid += 1
        for _ in range(12):
            kind, bad = damage(rng, png, prev_png)
            cfg = cfgs[int(rng.integers(0, len(cfgs)))]
            desired = int(rng.choice([3, 4]))
            st_r, out_r, *_ = judge(bad, desired)
            try:
                st_m, out_m, *_ = M.emul_decode(bad, desired, cfg)
            except AssertionError as ex:
                print(f"EMULATOR ERROR {ex}: seed {seed} file {files} {w}x{h}x{c} flags {flags} kind {kind} cfg {cfg} desired {desired} reference {st_r}", flush=True)
                open(f"/tmp/emul_campaign_err_{seed}_{files}_{damaged}.png", "wb").write(bad)
                continue
            if st_m == M.UNDECIDED:  # left to the CPU decoder (fpng_decode.cpp), as the drop-in does: that one must agree with the reference then
                undecided += 1
                st_m, out_m, *_ = cpu_tier(bad, desired)
                out_m = np.asarray(out_m)[: len(out_r)] if st_m == 0 and st_r == 0 else out_m
            damaged += 1
            rejected += st_r != 0
            by_kind[kind] = by_kind.get(kind, 0) + 1
            ok = st_m == st_r and (st_r != 0 or np.array_equal(np.asarray(out_r)[: out_m.size], out_m))
            if not ok:
                print(f"MISMATCH damaged file: seed {seed} file {files} {w}x{h}x{c} flags {flags} kind {kind} cfg {cfg} desired {desired} reference {st_r} emulator {st_m}", flush=True)
                open(f"/tmp/emul_campaign_fail_{seed}_{files}_{damaged}.png", "wb").write(bad)
        # token-level edits (tests/token_mutator.py): valid code streams that bend or break the decoder's semantic rules
        if w * h * c <= 60000 and M.plan(png)[1] == 0:
            ts = TM.Stream(png, M.plan)
            for _ in range(6):
                T, name = TM.mutate(ts, rng)
                if name == "none":
                    continue
                for bal in (False, True):
                    f = ts.write(TM.balanced(ts, T, rng) if bal else T)
                    if f is None:
                        continue
                    desired = int(rng.choice([3, 4]))
                    cfg = cfgs[int(rng.integers(0, len(cfgs)))]
                    st_r, out_r, *_ = judge(f, desired)
                    st_c, out_c, *_ = cpu_tier(f, desired)
                    try:
                        st_m, out_m, *_ = M.emul_decode(f, desired, cfg)
                    except AssertionError as ex:
                        print(f"EMULATOR ERROR {ex}: seed {seed} file {files} {w}x{h}x{c} flags {flags} edit {name} cfg {cfg} desired {desired} reference {st_r}", flush=True)
                        open(f"/tmp/emul_campaign_err_{seed}_{files}_{edits}.png", "wb").write(f)
                        continue
                    if st_m == M.UNDECIDED:
                        edits_undecided += 1
                        st_m, out_m = st_c, out_c
                    edits += 1
                    edits_accepted += st_r == 0
                    ok = all(st == st_r and (st_r != 0 or np.array_equal(np.asarray(out_r)[: o.size], o)) for st, o in ((st_m, out_m), (st_c, out_c)))
                    if not ok:
                        print(f"MISMATCH edited tokens: seed {seed} file {files} {w}x{h}x{c} flags {flags} edit {name} balanced {bal} cfg {cfg} desired {desired} reference {st_r} emulator {st_m} cpu tier {st_c}", flush=True)
                        open(f"/tmp/emul_campaign_fail_{seed}_{files}_e{edits}.png", "wb").write(f)
        # the tokens (as they are, edited, with the reserved length symbols put in) under other dynamic Huffman tables (tests/header_mutator.py)
        if w * h * c <= 60000 and M.plan(png)[1] == 0:
            ts = TM.Stream(png, M.plan)
            for k in range(6):
                toks = None
                if k >= 4:
                    toks = HM.with_reserved_symbols(ts, rng)
                elif k >= 2:
                    toks, _ = TM.mutate(ts, rng)
                    if rng.random() < 0.5:
                        toks = TM.balanced(ts, toks, rng)
                name, f = HM.reencode(ts, rng, toks)
                desired = int(rng.choice([3, 4]))
                st_r, out_r, *_ = judge(f, desired)
                st_c, out_c, *_ = cpu_tier(f, desired)
                try:
                    st_m, out_m, *_ = M.emul_decode(f, desired, cfgs[int(rng.integers(0, len(cfgs)))])
                except AssertionError as ex:
                    print(f"EMULATOR ERROR {ex}: seed {seed} file {files} {w}x{h}x{c} table edit {name}", flush=True)
                    open(f"/tmp/emul_campaign_err_{seed}_{files}_t{tables}.png", "wb").write(f)
                    continue
                if st_m == M.UNDECIDED:
                    st_m, out_m = st_c, out_c
                tables += 1
                tables_accepted += st_r == 0
                ok = all(st == st_r and (st_r != 0 or np.array_equal(np.asarray(out_r)[: o.size], o)) for st, o in ((st_m, out_m), (st_c, out_c)))
                if not ok:
                    print(f"MISMATCH other table: seed {seed} file {files} {w}x{h}x{c} {name} reserved symbols {k >= 4} desired {desired} reference {st_r} emulator {st_m} cpu tier {st_c}", flush=True)
                    open(f"/tmp/emul_campaign_fail_{seed}_{files}_t{tables}.png", "wb").write(f)
        # chunk- and block-level edits with good CRCs (tests/container_mutator.py): the container walk, zlib header, stored-block layout,
        # dynamic header -- fpng_get_info and fpng_decode_memory of the reference judge the drop-in's functions and the GPU decoder's host side
        for png2 in (png, oracle().encode(img, w, h, c, 2)) if w * h * c <= 200000 else ():
            for _ in range(6):
                name, f = CM.mutate(png2, rng)
                if name == "none":
                    continue
                b = np.frombuffer(f, dtype=np.uint8)
                gw, gh, gc = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
                gst = ref().L.ref_get_info(b.ctypes.data, b.size, C.byref(gw), C.byref(gh), C.byref(gc))
                mine = dropin.get_info(f)
                desired = int(rng.choice([3, 4]))
                st_r, out_r, *dr = judge(f, desired)
                st_c, out_c, *dc = cpu_tier(f, desired)
                try:
                    st_m, out_m, *_ = M.emul_decode(f, desired, cfgs[int(rng.integers(0, len(cfgs)))])
                except AssertionError as ex:
                    print(f"EMULATOR ERROR {ex}: seed {seed} file {files} {w}x{h}x{c} container edit {name}", flush=True)
                    open(f"/tmp/emul_campaign_err_{seed}_{files}_c{containers}.png", "wb").write(f)
                    continue
                if st_m == M.UNDECIDED:
                    st_m, out_m = st_c, out_c
                containers += 1
                containers_accepted += st_r == 0
                ok = mine[0] == gst and (gst != 0 or mine[1:] == (gw.value, gh.value, gc.value)) and st_c == st_r and st_m == st_r and (
                    st_r != 0 or (dr == dc and np.array_equal(out_r, out_c) and np.array_equal(np.asarray(out_r)[: out_m.size], out_m)))
                if not ok:
                    print(f"MISMATCH container edit: seed {seed} file {files} {w}x{h}x{c} edit {name} desired {desired} reference {st_r} / info {gst} cpu tier {st_c} / info {mine[0]} emulator {st_m}", flush=True)
                    open(f"/tmp/emul_campaign_fail_{seed}_{files}_c{containers}.png", "wb").write(f)
        prev_png = png
    print(f"seed {seed}: {tables} files under other Huffman tables ({tables_accepted} accepted by the reference)", flush=True)
    print(f"seed {seed}: {containers} files with edited containers / blocks ({containers_accepted} accepted by the reference)", flush=True)
    print(f"seed {seed}: {edits} files with edited tokens ({edits_accepted} accepted by the reference, {edits_undecided} left to the CPU decoder by the kernels' logic)", flush=True)
    print(f"seed {seed}: {files} files, {valid} decodes of valid files, {damaged} damaged copies ({rejected} rejected by the reference, {undecided} left undecided), by kind {dict(sorted(by_kind.items()))}", flush=True)


if __name__ == "__main__":
    main()
