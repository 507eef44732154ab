"""Generate tests/golden/batches.json from the UNMODIFIED reference (oracle/_ref/libfpng_ref.so): size + sha256 of
every image of the BASELINE.json batch configurations, so that the -m gpu tests can check whole submissions at
full scale without a CPU encode on the GPU box.

Run in the dev container (where /root/reference exists):   python oracle/make_golden_batches.py [set ...]
(with set names only those are regenerated, the others are kept from the existing file)

  c3   256 x 1920x1080x3 `grad`, seed 12345+i, flags 0          (BASELINE config 3)
  c5   128 x 3840x2160x4 `grad`, seed 12345+i, flags 1          (1/8 of BASELINE config 5)
  c4   16384x16384x4 `grad`, seed 777, flags 0 and 1            (BASELINE config 4; flags 1 = the 2-pass row-band case of config 5's wording)
  bench  8 x 7680x4320x4 `grad`, seed 12345+i, flags 0 and 1    (bench.py's default step)
  bench_noise  8 x 7680x4320x4 `noise`, flags 0                 (bench.py --kind noise: the stored outcome)
  bench_4k     16 x 3840x2160x4 `grad`, flags 0                 (bench.py --workload 4k --batch 16)
  bench_512    1024 x 512x512x3 `grad`, flags 0                 (bench.py --workload 512 --batch 1024; compact form)
  bench_16k    16384x16384x4 `grad`, seed 12345, flags 0        (bench.py --workload 16k --batch 1)

Sets of more than 300 images are stored compactly: "sha256_all" = sha256 over the concatenated per-image hex digests.
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SETS = {
    "c3": dict(w=1920, h=1080, c=3, kind="grad", n=256, seed0=12345, flags=[0]),
    "c5": dict(w=3840, h=2160, c=4, kind="grad", n=128, seed0=12345, flags=[1]),
    "c4": dict(w=16384, h=16384, c=4, kind="grad", n=1, seed0=777, flags=[0, 1]),
    "bench": dict(w=7680, h=4320, c=4, kind="grad", n=8, seed0=12345, flags=[0, 1]),
    "bench_noise": dict(w=7680, h=4320, c=4, kind="noise", n=8, seed0=12345, flags=[0]),
    "bench_4k": dict(w=3840, h=2160, c=4, kind="grad", n=16, seed0=12345, flags=[0]),
    "bench_512": dict(w=512, h=512, c=3, kind="grad", n=1024, seed0=12345, flags=[0]),
    "bench_16k": dict(w=16384, h=16384, c=4, kind="grad", n=1, seed0=12345, flags=[0]),
}


def one(task):
    name, i, fl = task
    from cpu_ref import ref
    import fpng_amd
    s = SETS[name]
    img = fpng_amd.synth_image(s["kind"], s["w"], s["h"], s["c"], seed=s["seed0"] + i)
    png = ref().encode(img, s["w"], s["h"], s["c"], fl)
    return name, i, fl, len(png), hashlib.sha256(png).hexdigest()


def main():
    only = sys.argv[1:]
    path = os.path.join(ROOT, "tests", "golden", "batches.json")
    out = {}
    if only:
        with open(path) as f:
            out = json.load(f)
    sets = {k: v for k, v in SETS.items() if not only or k in only}
    tasks = [(name, i, fl) for name, s in sets.items() for fl in s["flags"] for i in range(s["n"])]
    tasks.sort(key=lambda t: -SETS[t[0]]["w"] * SETS[t[0]]["h"])
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(one, tasks, chunksize=1)
    for name, s in sets.items():
        e = {k: s[k] for k in ("w", "h", "c", "kind", "n", "seed0")}
        e["flags"] = {}
        for fl in s["flags"]:
            rows = sorted((i, size, sha) for (nm, i, f, size, sha) in res if nm == name and f == fl)
            if len(rows) > 300:
                e["flags"][str(fl)] = {"total_size": sum(r[1] for r in rows),
                                       "sha256_all": hashlib.sha256("".join(r[2] for r in rows).encode()).hexdigest()}
            else:
                e["flags"][str(fl)] = {"sizes": [r[1] for r in rows], "sha256": [r[2] for r in rows]}
        out[name] = e
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote tests/golden/batches.json:", {k: len(v["flags"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
