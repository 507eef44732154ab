#!/usr/bin/env python
"""bench.py -- throughput of the fpng encode hot path (and of the GPU decoder on the encoder's output) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 8k|4k|1080p] [--batch B] [--flags F]

A "step" is one submission of the hot path (fpng_encode_image_to_memory semantics, 1-pass) over a
batch of B synthetic device-resident images; inputs and outputs live in HBM for the whole timed
region (no PCIe).  Default workload: the configuration BASELINE.json's metric is quoted on --
7680x4320 RGBA `grad` frames, B distinct frames per step so the working set
(B x 133 MB in + PNG out) is far larger than the 256 MiB Infinity Cache.

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) every rank encodes its own
batch (images are independent objects: no data-path collective), time is max over ranks,
value = all ranks' pixels / that time, scaling = weak.

Rank 0 prints ONE JSON line, including
  roofline     : dominant kernel's algorithmic bytes / its HIP-event-measured duration vs 8 TB/s
  cpu_baseline : the reference's own SSE4.1 encoder (oracle/_ref) timed on one host core on a
                 bounded sample of the same workload (falls back to the C port in oracle/)
"""
import argparse
import re
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {"8k": (7680, 4320, 4), "4k": (3840, 2160, 4), "1080p": (1920, 1080, 3), "1080p4": (1920, 1080, 4), "512": (512, 512, 3), "16k": (16384, 16384, 4)}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # ... and what streaming kernels reach (same guide): roofline.frac_of_achievable
PASS_NAME = {0: "1-pass", 1: "2-pass (FPNG_ENCODE_SLOWER)", 2: "stored (FPNG_FORCE_UNCOMPRESSED)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="8k")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out-sets", type=int, default=8, help="sets of output buffers the submissions cycle through (one per submission that can be in flight)")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--kind", default="grad")
    ap.add_argument("--prewarm", type=int, default=60,
                    help="untimed steps before the W warmup steps: throughput needs ~40 back-to-back steps (20-50 ms of "
                         "sustained load) to settle, a cold 20-step run reads 15-25 %% low (DESIGN.md section 6)")
    ap.add_argument("--mode", default="batch", choices=["batch", "rowband", "decode", "nodeimage"],
                    help="batch: every rank encodes its own images (BASELINE configs 2/3/5, the headline; its line also carries a "
                         "`decode` object); rowband: ONE image sharded by rows over the ranks, one IDAT, windows gathered to rank 0 "
                         "over RCCL (BASELINE config 4); decode: the line is about the GPU decoder (the files the encoder just wrote, "
                         "still in device memory, back to pixels in device memory)")
    ap.add_argument("--pipelines", type=int, default=0, help="--mode nodeimage: devices of the node (default: every visible GPU; on a one-GPU box "
                    "device 0 listed this many times)")
    ap.add_argument("--decode-steps", type=int, default=20, help="timed decode steps per region (the `decode` object / --mode decode)")
    ap.add_argument("--regions", type=int, default=3,
                    help="timed regions per run, each EXACTLY --steps steps between two barriers; value = the median region, "
                         "`runs` / `spread` report all of them (a single 10 ms region cannot show a 5 %% change)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--end-to-end-only", default=None, choices=["cabi", "dropin"], help="(internal) print that part of the end_to_end object for --workload WxHxC and exit")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch check without a GPU: the ranks --gpus N asks for are started (gloo), meet at the barriers of a timed region "
                         "that holds no kernels, and rank 0 prints the one JSON line with n_gpus = the ranks that actually ran (tests/test_bench_launch.py)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become N ranks (one per GPU) under torch.distributed.run, the
    command line unchanged.  The parent makes no HIP call and prints nothing of its own: the one JSON line is rank 0's."""
    import socket
    import subprocess
    if not args.dry_launch:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); a run over fewer GPUs than asked for would not be the run asked for")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def dry_launch(args, rank, world):
    """The launch path of a scaling run with everything GPU-side left out: rendezvous (gloo), the barriers of one timed region, the
    max over ranks, one line from rank 0.  What it shows: `--gpus N` really is N processes, and `n_gpus` counts processes, not flags."""
    import torch.distributed as dist
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    def barrier():
        if dist.is_initialized():
            dist.barrier()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass  # (a step of the real run: enc.submit(...))
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ranks = torch.ones(1, dtype=torch.int64)
    if dist.is_initialized():
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
    if rank == 0:
        _RESULT.append(json.dumps({"metric": "launch check only (--dry-launch): no kernels ran", "value": None, "unit": "MP/s", "n_gpus": int(ranks.item()),
                                   "gpus_asked_for": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "dry_launch": True,
                                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                                   "config": {"workload": "none", "mode": args.mode, "backend": "gloo" if dist.is_initialized() else "none"}}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def workload_tag(args):
    """Name of this command in profiles/: r03_<tag>_pmc_traffic.txt etc. (tools/gpu_profile_round.sh <round>_<tag> <args>)."""
    defaults = {"8k": 8, "4k": 16, "1080p": 256, "512": 1024, "16k": 1}
    tag = args.workload + ("" if defaults.get(args.workload, args.batch) == args.batch else f"_b{args.batch}")
    if args.flags:
        tag += {1: "_2pass", 2: "_stored"}.get(args.flags, f"_f{args.flags}")
    if args.kind != "grad":
        tag += "_" + args.kind
    return tag


def _pmc_summaries(tag):
    """profiles/rNN[x]_<tag>_pmc_traffic.txt, newest first -- the WHOLE basename is matched: `8k` must not pick up
    `decode_8k` (round 4's driver line read the decoder's file for the encode chain and printed a rate above the HBM peak).
    Round 1-2 files without a workload tag (rNN_x_pmc_traffic.txt) are the default 8k command."""
    import re
    pat = re.compile(r"r\d\d[a-z]?_" + re.escape(tag) + r"_pmc_traffic\.txt")
    old = re.compile(r"r\d\d_[a-z]_pmc_traffic\.txt")
    d = os.path.join(ROOT, "profiles")
    names = [n for n in (os.listdir(d) if os.path.isdir(d) else []) if pat.fullmatch(n) or (tag == "8k" and old.fullmatch(n))]
    return [os.path.join(d, n) for n in sorted(names, reverse=True)]


def committed_traffic(kernel, tag):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary of THIS command (profiles/rNN_<tag>_pmc_traffic.txt,
    produced by tools/gpu_profile_round.sh + tools/summarize_profile.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    runs).  PMC counters cannot be collected from inside the timed process."""
    for path in _pmc_summaries(tag):
        for line in open(path):
            f = line.split()
            if len(f) in (5, 7) and f[0] == kernel:  # (round 6 on: + launches per step, megabytes per step; the first five columns are ONE launch's)
                import hashlib
                # (the file's content hash goes along: a summary from another tree cannot pass for this one's unnoticed)
                # (a kernel that is launched several times per step -- dec_sync_kernel: round 0 and two border rounds -- is timed as ONE
                #  phase by the library: its bytes are the step's too, the last column)
                mb = float(f[6]) if len(f) == 7 else float(f[2]) + float(f[4])
                return int(mb * 1e6), os.path.relpath(path, ROOT) + "#sha256=" + hashlib.sha256(open(path, "rb").read()).hexdigest()[:12]
    return None, None


def committed_chain_traffic(tag):
    """HBM-side bytes of ALL kernels of one step, from the same committed PMC summary ("total HBM-side traffic per launch")."""
    import re
    for path in _pmc_summaries(tag):
        m = re.search(r"total HBM-side traffic per launch:\s*([0-9.]+) MB", open(path).read())
        if m:
            return int(float(m.group(1)) * 1e6)
    return None


def scope_name(world):
    return "one GPU" if world == 1 else f"whole job, {world} GPUs"


def check_rates(obj, path="line", errors=None):
    """No rate this script prints may exceed the HBM peak of the GPUs it ran on: such a number is a reporting bug.  The field is
    set to None and named in the returned list (the line then carries `reporting_errors`; tests/test_abi.py holds the rule)."""
    errors = [] if errors is None else errors
    if isinstance(obj, dict):
        peak = obj.get("peak", HBM_PEAK_GBS) if obj.get("unit") == "GB/s" else HBM_PEAK_GBS
        for k, v in list(obj.items()):
            if isinstance(v, (dict, list)):
                check_rates(v, f"{path}.{k}", errors)
            elif isinstance(v, (int, float)) and (k == "achieved" or k.endswith("_GBs")) and obj.get("bound", "hbm") == "hbm" and v > peak:
                errors.append(f"{path}.{k} = {v} GB/s is above the HBM peak ({peak} GB/s)")
                obj[k] = None
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            check_rates(v, f"{path}[{i}]", errors)
    return errors


def verify_outputs(args, rank, w, h, c, outs, sizes):
    """sha256 of EVERY image of the last timed submission (seeds 12345+i on rank 0) vs the golden vectors produced by the
    unmodified reference (tests/golden/batches.json holds whole batches, kat.json single images with seed 12345).
    Returns the number of images checked (all of them must have a golden vector), or None when there is none for this input."""
    import hashlib
    if rank != 0 or args.flags not in (0, 1, 2):
        return None
    want = {}
    want_all = None  # compact sets: (number of images, sha256 over the concatenated per-image hex digests)
    try:
        with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
            for e in json.load(f):
                if (e["w"], e["h"], e["c"], e["kind"]) == (w, h, c, args.kind) and str(args.flags) in e["flags"]:
                    want[0] = e["flags"][str(args.flags)]["sha256"]
        with open(os.path.join(ROOT, "tests", "golden", "batches.json")) as f:
            for e in json.load(f).values():
                if (e["w"], e["h"], e["c"], e["kind"], e["seed0"]) == (w, h, c, args.kind, 12345) and str(args.flags) in e["flags"]:
                    g = e["flags"][str(args.flags)]
                    if "sha256_all" in g:
                        want_all = (e["n"], g["sha256_all"])
                    for i, sha in enumerate(g.get("sha256", [])):
                        want[i] = sha
    except OSError:
        return None
    n = 0
    digests = []
    for i, (out, size) in enumerate(zip(outs, sizes)):
        if i not in want and not (want_all and len(outs) == want_all[0]):
            continue
        got = hashlib.sha256(out[:size].cpu().numpy().tobytes()).hexdigest()
        digests.append(got)
        if i in want:
            if got != want[i]:
                raise SystemExit(f"bench.py: PARITY FAILURE, image {i} sha256 {got} != reference {want[i]}")
            n += 1
    if want_all and len(outs) == want_all[0]:
        if hashlib.sha256("".join(digests).encode()).hexdigest() != want_all[1]:
            raise SystemExit(f"bench.py: PARITY FAILURE, the {len(outs)} files' digests do not hash to the reference's {want_all[1]}")
        n = len(outs)
    return n or None


def _cpu_worker(task):
    w, h, c, kind, flags, reps, seed = task
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ref
    import fpng_amd
    img = fpng_amd.synth_image(kind, w, h, c, seed=seed)
    if cpu_ref.have_ref():
        return cpu_ref.ref().time_encode(img, w, h, c, flags, reps)[0]
    o = cpu_ref.oracle()
    t0 = time.perf_counter()
    o.encode(img, w, h, c, flags)
    return time.perf_counter() - t0


def end_to_end(enc_device, w, h, c, kind, flags):
    """The host-pixel paths in a FRESH process each: a user of the reference's API has no other GPU work in its process."""
    import subprocess
    res = {}
    for part in ("cabi", "dropin"):  # (one streaming encoder per process: the C ABI's, then the drop-in's)
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--end-to-end-only", part, "--workload", f"{w}x{h}x{c}", "--kind", kind,
                                  "--flags", str(flags), "--device", str(enc_device)], capture_output=True, text=True, timeout=600)
            got = None
            for ln in reversed(out.stdout.strip().splitlines()):
                if ln.startswith("{"):
                    got = json.loads(ln)
                    break
            res.update(got if got is not None else {f"{part}_error": (out.stderr or out.stdout)[-200:]})
        except Exception as e:
            res[f"{part}_error"] = str(e)[:200]
    return res


def end_to_end_here(enc_device, w, h, c, kind, flags, part):
    """The drop-in's view (SURVEY 8d timing 2): host pixels in, host PNG out, PCIe inclusive -- one blocking call per frame
    (what fpng::fpng_encode_image_to_memory does) and the many-frames form whose copies overlap.  Never `value`."""
    import fpng_amd
    n = 6
    imgs = [fpng_amd.synth_image(kind, w, h, c, seed=12345 + i) for i in range(n)]
    mp = w * h / 1e6
    if part == "dropin":
        # through libfpng.so itself: fpng::fpng_encode_image_to_memory() into one reused std::vector, timed in C++ like the
        # reference's harness does (fpng_test.cpp:1198-1209; SURVEY 8d timing 2 "incl. vector resize"), then a fresh vector per call
        out = {}
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import dropin
            t, _ = dropin.time_encode(imgs[0], w, h, c, flags, reps=8, reuse=True)
            tf, _ = dropin.time_encode(imgs[0], w, h, c, flags, reps=3, reuse=False)
            out["dropin_ms"] = round(t * 1e3, 3)
            out["dropin_MPs"] = round(mp / t, 1)
            out["dropin_fresh_vector_ms"] = round(tf * 1e3, 3)
            # ... and back: fpng::fpng_decode_memory() of that file into a reused std::vector (images of 256K pixels and more go
            # through the GPU decoder: upload, decode, download)
            png = dropin.encode(imgs[0], w, h, c, flags)
            td = dropin.time_decode(png, c, reps=4)
            out["dropin_decode_ms"] = round(td * 1e3, 3)
            out["dropin_decode_MPs"] = round(mp / td, 1)
            out["dropin_decode_on_gpu"] = bool(dropin.gpu_decodes() > 0)
            # images under 256K pixels stay on the drop-in's CPU decoder (fpng_decode.cpp): one host core, next to the reference's
            sw, sh = 512, 500
            simg = fpng_amd.synth_image(kind, sw, sh, c, seed=12345)
            spng = dropin.encode(simg, sw, sh, c, flags)
            n0 = dropin.gpu_decodes()
            ts = min(dropin.time_decode(spng, c, reps=20) for _ in range(3))
            out["dropin_small_decode"] = {"image": f"{sw}x{sh}x{c}", "MPs": round(sw * sh / ts / 1e6, 1), "on_gpu": bool(dropin.gpu_decodes() > n0)}
            import cpu_ref
            if cpu_ref.have_ref():
                tr = min(cpu_ref.ref().time_decode(spng, c, 20) for _ in range(3))
                out["dropin_small_decode"]["reference_MPs"] = round(sw * sh / tr / 1e6, 1)
        except Exception as e:  # (needs g++ for the test shim)
            out["dropin_error"] = str(e)[:80]
        return out
    outs = [np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8) for _ in range(n)]
    enc = fpng_amd.Encoder(device=enc_device, stream="own")
    best1, bestn = 1e30, 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        enc.encode_host_into(imgs[0], w, h, c, outs[0], flags)
        best1 = min(best1, time.perf_counter() - t0)
        t0 = time.perf_counter()
        enc.encode_host_batch(imgs, flags, outs=outs)
        bestn = min(bestn, (time.perf_counter() - t0) / n)
    # the same frame in PAGE-LOCKED memory (fpng_amd_pin_host_memory): streamed through the GPU in row bands, upload | encode |
    # download overlapped
    bestp = None
    try:
        fpng_amd.pin_host_memory(imgs[0])
        bestp = 1e30
        for _ in range(4):
            t0 = time.perf_counter()
            enc.encode_host_into(imgs[0], w, h, c, outs[0], flags)
            bestp = min(bestp, time.perf_counter() - t0)
        fpng_amd.unpin_host_memory(imgs[0])
    except Exception:
        bestp = None
    enc.close()
    out = {"single_call_ms": round(best1 * 1e3, 3), "single_call_MPs": round(mp / best1, 1), "frames_per_batch_call": n,
           "batch_ms_per_frame": round(bestn * 1e3, 3), "batch_MPs": round(mp / bestn, 1),
           "note": "host pixels -> host PNG through the C ABI the fpng:: drop-in uses, pageable memory, PCIe inclusive"}
    if bestp:
        out["single_call_page_locked_ms"] = round(bestp * 1e3, 3)
        out["single_call_page_locked_MPs"] = round(mp / bestp, 1)
    return out


def cpu_baseline(w, h, c, kind, flags, reps):
    """The reference's CPU path on ONE host core (rank 0, N=1 only).  Checker-side code: this is the
    only place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ref
    import fpng_amd
    img = fpng_amd.synth_image(kind, w, h, c)
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except Exception:
        pass
    mp = w * h / 1e6
    if cpu_ref.have_ref():
        r = cpu_ref.ref()
        secs, size = r.time_encode(img, w, h, c, flags, reps)
        out = {"value": round(mp / secs, 2), "unit": "MP/s", "cores": 1, "kind": "reference",
               "sample": f"1 image {w}x{h}x{c} {kind}, best of {reps}, fpng.cpp SSE4.1+PCLMUL build (sse41={r.L.ref_supports_sse41()})",
               "png_bytes": size, "value_MiPs": round(mp / secs * 1e6 / 2 ** 20, 2)}
        # whole-node figure: one process per host core, one image each
        try:
            import multiprocessing as mp_
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
            cores = os.cpu_count() or cores
            with mp_.get_context("spawn").Pool(cores) as pool:
                ts = pool.map(_cpu_worker, [(w, h, c, kind, flags, 2, 12345 + i) for i in range(cores)])
            out["all_cores"] = {"processes": cores, "value": round(cores * mp / max(ts), 1), "unit": "MP/s",
                                "sample": f"{cores} processes, one {w}x{h}x{c} image each, best of 2, slowest process counts"}
        except Exception as e:  # the single-core figure is the contract; the node figure is extra
            out["all_cores"] = {"error": str(e)[:100]}
        return out
    o = cpu_ref.oracle()
    best = 1e30
    for _ in range(max(1, reps // 2)):
        t0 = time.perf_counter()
        png = o.encode(img, w, h, c, flags)
        best = min(best, time.perf_counter() - t0)
    return {"value": round(mp / best, 2), "unit": "MP/s", "cores": 1, "kind": "port",
            "sample": f"1 image {w}x{h}x{c} {kind}, best of {max(1, reps // 2)}, scalar C port", "png_bytes": len(png)}


def decode_cpu_baseline(png, w, h, c):
    """The reference's own decoder (oracle/_ref, fpng_decode_memory into a reused vector, best of 3: the way
    reference/src/fpng_test.cpp:1236-1273 times it) on ONE host core; without its build the drop-in's CPU decoder."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ref
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
    except Exception:
        pass
    mp = w * h / 1e6
    try:
        if cpu_ref.have_ref():
            secs = cpu_ref.ref().time_decode(png, c, 3)
            return {"value": round(mp / secs, 2), "unit": "MP/s", "cores": 1, "kind": "reference",
                    "sample": f"1 file {w}x{h}x{c} ({len(png)} bytes), fpng_decode_memory best of 3, fpng.cpp SSE4.1 build"}
        os.environ["FPNG_AMD_DECODE_CPU"] = "1"
        import dropin
        secs = dropin.time_decode(png, c, reps=3)
        return {"value": round(mp / secs, 2), "unit": "MP/s", "cores": 1, "kind": "port",
                "sample": f"1 file {w}x{h}x{c} ({len(png)} bytes), the drop-in's CPU decoder (fpng_decode.cpp), best of 3"}
    except Exception as e:
        return {"error": str(e)[:120]}
    finally:
        os.environ.pop("FPNG_AMD_DECODE_CPU", None)
        try:
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
        except Exception:
            pass


def decode_bench(args, enc, imgs, pngs, w, h, c, rank, world, barrier, all_max, with_cpu):
    """A decode step = ONE fpng_amd_decode_batch_device() call over the B files the encoder wrote (device-resident; only a head and
    a tail of each file visit the host for the container walk), pixels back into device memory; the call returns when they are
    there, so the K steps of a region run one after another.  Roofline: algorithmic bytes = the files read once + the pixels
    written once, against the kernel with the longest HIP-event time; parity: the pixels must equal the frames that were encoded."""
    B = len(imgs)
    dims = [(w, h)] * B
    outs = [torch.empty(w * h * c, dtype=torch.uint8, device=imgs[0].device) for _ in range(B)]
    # (the call's descriptor array is built once, as make_batch() does for the encode steps: filling it takes Python ~3 us a file --
    #  for 1024 files of 512 x 512 more than the GPU needs for them; a C caller has no such cost)
    db = enc.make_decode_batch(pngs, c, dims, outs)
    for _ in range(3):
        enc.decode_device(db, results=False)
    K = max(1, args.decode_steps)
    runs = []
    for _ in range(max(1, args.regions)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(K):
            enc.decode_device(db, results=False)
        barrier()
        runs.append(all_max(time.perf_counter() - t0))
    elapsed = sorted(runs)[len(runs) // 2]
    got = db.results()
    if not all(st == 0 for st, _, _ in got):
        raise SystemExit(f"bench.py: decode status {[st for st, _, _ in got]}")
    for i, ((st, px, _), t) in enumerate(zip(got, imgs)):
        if not torch.equal(px, t):
            raise SystemExit(f"bench.py: DECODE PARITY FAILURE, file {i}: pixels differ from the encoded frame")
    enc.set_profiling(True)
    ph = {}
    reps = 5
    for _ in range(reps):
        enc.decode_device(db, results=False)
        for k, v in enc.last_decode_phase_ms().items():
            ph[k] = ph.get(k, 0.0) + v / reps
    enc.set_profiling(False)
    png_bytes = sum(int(p.numel()) for p in pngs)
    alg = png_bytes + B * w * h * c
    dom = max(ph, key=lambda k: ph[k])
    kernels_ms = sum(ph.values())
    value = world * B * w * h * K / elapsed / 1e6
    traffic, traffic_src = committed_traffic(f"dec_{dom}_kernel", "decode_" + workload_tag(args))
    out = {"metric": f"decode megapixels/sec ({scope_name(world)}), device-resident files -> device-resident pixels", "value": round(value, 1), "unit": "MP/s",
           "steps": K, "ms_per_step": round(elapsed / K * 1e3, 4), "runs": [round(world * B * w * h * K / r / 1e6, 1) for r in runs],
           "parity_checked": True, "parity_images": B, "png_bytes_per_step_per_gpu": png_bytes,
           "roofline": {"bound": "hbm", "kernel": f"dec_{dom}_kernel", "achieved": round(alg / (ph[dom] / 1e3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (ph[dom] / 1e3) / 1e9 / HBM_PEAK_GBS, 4), "frac_of_achievable": round(alg / (ph[dom] / 1e3) / 1e9 / HBM_ACHIEVABLE_GBS, 4),
                        "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes_per_launch": alg, "kernel_ms": round(ph[dom], 4), "all_kernels_ms": round(kernels_ms, 4),
                        "pipeline_frac": round(alg / (kernels_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4), "step_frac": round(alg / (elapsed / K) / 1e9 / HBM_PEAK_GBS, 4),
                        "phase_ms": {k: round(v, 4) for k, v in ph.items()}}}
    if with_cpu:
        out["cpu_baseline"] = decode_cpu_baseline(bytes(pngs[0].cpu().numpy()), w, h, c)
        if "value" in out["cpu_baseline"]:
            out["speedup_vs_cpu_1core"] = round(value / out["cpu_baseline"]["value"], 1)
    return out


def nodeimage(args, w, h, c):
    """--mode nodeimage: ONE host-resident image through fpng_amd_node_encode_host_image (row bands dealt to the node's devices, every
    device moving its band up and its window down over its own link).  PCIe inclusive -- a host path, never the headline `value`
    of the default mode.  ONE process drives the first --gpus devices (n_gpus = that count; fewer visible devices is an error); --pipelines
    bands are dealt to them in turn (default: one per device, eight through the one link of a one-GPU run).
    This is BASELINE config 4's scaling line: the file must end up in ONE memory, and a host-resident image reaches N GPUs over N PCIe
    links (model: 28 ms on one link -> 3.9 ms on eight), whereas --mode rowband (rows already in the GPUs' memories,
    windows gathered over xGMI into rank 0's) is by the same model SLOWER on eight GPUs than on one (1.3 vs 0.50 ms)."""
    import hashlib
    import fpng_amd
    ngpu = torch.cuda.device_count()
    if ngpu < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {ngpu} GPU(s)")
    n_pipes = max(1, args.pipelines or (args.gpus if args.gpus > 1 else 8))
    devices = [i % args.gpus for i in range(n_pipes)]
    img = fpng_amd.synth_image(args.kind, w, h, c, seed=777 if args.workload == "16k" else 12345)
    node = fpng_amd.Node(devices)
    out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)  # (reused from call to call, like the harness's vector)
    for _ in range(max(1, args.warmup)):
        n = node.encode_host_image(img, w, h, c, args.flags, out)
    times = []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        n = node.encode_host_image(img, w, h, c, args.flags, out)
        times.append(time.perf_counter() - t0)
    node.close()
    png = out[:n].tobytes()
    best, med = min(times), sorted(times)[len(times) // 2]
    parity = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "batches.json")) as f:
            g = json.load(f)["c4"]
        if (w, h, c, args.kind) == (g["w"], g["h"], g["c"], g["kind"]) and str(args.flags) in g["flags"]:
            if hashlib.sha256(png).hexdigest() != g["flags"][str(args.flags)]["sha256"][0]:
                raise SystemExit("bench.py: PARITY FAILURE, node image file differs from the reference's")
            parity = True
    except OSError:
        pass
    bytes_moved = w * h * c + len(png)
    _RESULT.append(json.dumps({
        "metric": f"encode megapixels/sec, ONE host-resident image over the node's devices, {PASS_NAME.get(args.flags)}, PCIe inclusive",
        "value": round(w * h / med / 1e6, 1), "unit": "MP/s", "n_gpus": len(set(devices)), "pipelines": len(devices), "steps": args.steps, "warmup": max(1, args.warmup),
        "ms_per_step": round(med * 1e3, 3), "best_ms": round(best * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic", "parity_checked": parity,
        "config": {"workload": f"ONE {w}x{h} {'RGBA' if c == 4 else 'RGB'} '{args.kind}' image in host memory -> one fpng file in host memory, flags={args.flags}",
                   "devices": devices, "png_bytes": len(png), "host_bytes_moved": bytes_moved,
                   "parallelism": "row bands, one per listed device; band up / window down over the device's own PCIe link; 64-byte records meet on the host",
                   "baseline_config": "4 (one 16384x16384 image over the node): the line to scale with --gpus; --mode rowband gathers the windows over xGMI "
                                      "into one GPU's memory and is slower on 8 GPUs than on 1 by its own cost model (DESIGN.md section 5)"},
        "roofline": {"bound": "pcie", "note": "host path: bytes over the links / time", "achieved": round(bytes_moved / med / 1e9, 1), "unit": "GB/s",
                     "peak": None, "frac": None, "traffic": None}}))


def rowband(args, rank, local_rank, world, distributed, dev, w, h, c):
    """One image, rows sharded over the ranks (SURVEY 8e / BASELINE config 4).  A step = the whole exchange: band encode,
    all_gather of the band records, placement at the band's bit position, windows to rank 0, wrap.  Steps cannot be
    pipelined (each one has the host-visible exchange in the middle), so they are timed one after another."""
    import hashlib
    import torch.distributed as dist
    import fpng_amd
    from fpng_amd import sharded
    if not distributed:  # a one-rank group so that the same code path (incl. the nccl collectives) runs
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    img = fpng_amd.synth_image(args.kind, w, h, c, seed=777 if args.workload == "16k" else 12345)
    y0, y1 = sharded.split_rows(h, world)[rank]
    rows = torch.from_numpy(img[y0:y1]).to(dev)
    above = torch.from_numpy(img[y0 - 1]).to(dev) if y0 else None
    del img
    enc = fpng_amd.Encoder(device=local_rank)
    # the exchange runs behind the C ABI (fpng_amd_encode_image_sharded over the built-in RCCL transport); torch.distributed
    # only carries RCCL's 128-byte id to the ranks and the barriers around the timed region
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(sharded.CppRowSharded.rccl_unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, 0)
    sh = sharded.CppRowSharded(enc, rank, world, bytes(uid.cpu().numpy()), local_rank)
    out = torch.empty(fpng_amd.max_encoded_size(w, h, c) + 64, dtype=torch.uint8, device=dev) if rank == 0 else None

    def step():
        return sh.encode(rows, above, w, h, c, y0, y1, args.flags, 0, out)  # (returns when the file is complete on rank 0)

    for _ in range(max(1, args.warmup)):
        png = step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        png = step()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        parity = None
        try:
            with open(os.path.join(ROOT, "tests", "golden", "batches.json")) as f:
                g = json.load(f)["c4"]
            if (w, h, c, args.kind) == (g["w"], g["h"], g["c"], g["kind"]) and str(args.flags) in g["flags"]:
                got = hashlib.sha256(png.cpu().numpy().tobytes()).hexdigest()
                if got != g["flags"][str(args.flags)]["sha256"][0]:
                    raise SystemExit(f"bench.py: PARITY FAILURE, row-band file sha256 {got}")
                parity = True
        except OSError:
            pass
        png_bytes = int(png.numel())
        alg = w * h * c + png_bytes
        _RESULT.append(json.dumps({
            "metric": f"encode megapixels/sec ({scope_name(world)}), {PASS_NAME.get(args.flags, 'flags=%d' % args.flags)}, device-resident", "value": round(w * h * args.steps / elapsed / 1e6, 1),
            "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup), "prewarm": 0, "parity_checked": parity,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"ONE {w}x{h} {'RGBA' if c == 4 else 'RGB'} '{args.kind}' image as {world} row band(s), one IDAT, "
                                   f"flags={args.flags}, windows gathered to rank 0, bit-exact fpng PNG output", "width": w, "height": h,
                       "channels": c, "png_bytes": png_bytes, "parallelism": f"rows sharded over {world} GPU(s) behind the C ABI "
                       "(fpng_amd_encode_image_sharded, RCCL transport): all_gather of a 64-byte and a 16-byte record per rank "
                       "(+ all_reduce of 288 counters for 2-pass), windows sent to rank 0",
                       "expectation": "for rows that already live on several GPUs; by the cost model (DESIGN.md section 5) 8 ranks take ~1.3 ms where one GPU takes 0.50 ms, "
                                      "because 7 windows of 58.6 MB converge on rank 0 over one xGMI link each: BASELINE config 4's scaling line is --mode nodeimage"},
            "roofline": {"bound": "hbm", "kernel": "whole step (band encode + exchange + place + gather + wrap)",
                         "achieved": round(alg / (elapsed / args.steps) / 1e9, 1), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": round(alg / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 4), "traffic": None,
                         "algorithmic_bytes_per_launch": alg},
        }))
    sh.close()
    enc.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    if args.end_to_end_only:
        w, h, c = (int(v) for v in args.workload.split("x"))
        import fpng_amd  # noqa: F401  (as below: before the first HIP call)
        torch.cuda.set_device(args.device)
        print(json.dumps(end_to_end_here(args.device, w, h, c, args.kind, args.flags, args.end_to_end_only)), flush=True)
        return
    if args.workload not in WORKLOADS and not re.fullmatch(r"\d+x\d+x[34]", args.workload):
        raise SystemExit(f"--workload: one of {sorted(WORKLOADS)}, or WxHxC")
    distributed = "RANK" in os.environ  # launched by torch.distributed.run (also exercised with one rank)
    if args.gpus < 1:
        raise SystemExit("--gpus: at least 1")
    if args.gpus > 1 and not distributed and args.mode != "nodeimage":  # (nodeimage: ONE process drives --gpus devices)
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if distributed and world != args.gpus and args.mode != "nodeimage":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): n_gpus would not be what the command says")
    if args.dry_launch:
        return dry_launch(args, rank, world)
    import fpng_amd  # (before the first HIP call of the process: loading the library sets the runtime's hardware queues, csrc/api.cpp runtime_defaults())
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        # the first collective builds the communicator (tens of ms with the GPU idle): pay for it here, not in the barrier
        # that opens the timed region, where it would let the clocks drop right before the timed steps
        dist.barrier()
        torch.cuda.synchronize()
    dev = torch.device("cuda", local_rank)

    w, h, c = WORKLOADS[args.workload] if args.workload in WORKLOADS else tuple(int(v) for v in args.workload.split("x"))  # (a name, or WxHxC)
    if args.mode == "rowband":
        return rowband(args, rank, local_rank, world, distributed, dev, w, h, c)
    if args.mode == "nodeimage":
        return nodeimage(args, w, h, c)
    B = args.batch
    # B distinct frames per rank (seed varies per image and per rank)
    imgs = [torch.from_numpy(fpng_amd.synth_image(args.kind, w, h, c, seed=12345 + rank * 1000 + i)).to(dev) for i in range(B)]
    cap = fpng_amd.max_encoded_size(w, h, c) + 64
    # consecutive submissions overlap on the GPU (up to eight encoder lanes, four by default): each one in flight has its own set of output buffers
    n_sets = max(1, args.out_sets)
    out_sets = [[torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(B)] for _ in range(n_sets)]
    outs = out_sets[0]
    enc = fpng_amd.Encoder(device=local_rank, stream="own")
    rt = fpng_amd.runtime_info()

    batches = [enc.make_batch(imgs, o) for o in out_sets]  # descriptor arrays built once, like a capture pipeline would

    def step():
        enc.submit(imgs, outs, args.flags)
        return enc.finish(B)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup: W steps enqueued back to back like the timed ones (the first time two submissions overlap on the GPU
    # costs several ms once per process; with one finish per warmup step that would land in the timed region)
    # ... in TWO bursts with a finish in between: the burst after the first one that ran the lanes at full overlap is slow once more
    # (short rows, four lanes: one slot wait of 7-12 ms instead of 3.8; tools/region_diag.py) -- the timed regions are to be the third and later
    n_warm = args.prewarm + args.warmup
    for i in range(n_warm):
        enc.submit(batches[i % n_sets], None, args.flags)
        if i + 1 == n_warm // 2:
            res = enc.finish(B)
    if n_warm:
        res = enc.finish(B)
        # ... and one more untimed burst shaped like a timed region (barrier, K steps, finish): the first one of a process still runs 10-20 %
        # slow on the short-row workloads whatever the length of the warm-up in front of it (profiles/r05_hw_queues.txt, section 5)
        barrier()
        for i in range(args.steps):
            enc.submit(batches[i % n_sets], None, args.flags)
        res = enc.finish(B)
    # `regions` timed regions, each EXACTLY K steps enqueued back to back (the encoder pipelines submissions through a ring
    # of pinned slots) between a barrier + synchronize on both sides; per region the max over ranks counts.  The reported
    # value is the MEDIAN region; all of them are in `runs` (boxes and clocks wander by a few per cent within a run).
    runs = []
    own_runs = []  # per region: every rank's OWN time to its last result (before the closing barrier): a straggler shows here
    for _ in range(max(1, args.regions)):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            enc.submit(batches[i % n_sets], None, args.flags)
        res = enc.finish(B)
        own = time.perf_counter() - t0
        barrier()
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            g = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(g, torch.tensor([own], dtype=torch.float64, device=dev))
            own_runs.append([float(v.item()) for v in g])
        else:
            own_runs.append([own])
        runs.append(el)
    elapsed = sorted(runs)[len(runs) // 2]
    per_rank_ms = [round(v / args.steps * 1e3, 4) for v in own_runs[runs.index(elapsed)]]

    png_bytes = sum(r[0] for r in res)
    assert all(r[2] == 0 for r in res)
    # self-check: EVERY image of the last timed submission's output set against the reference's bytes
    # (tests/golden/batches.json / kat.json: sha256 of the unmodified reference encoder's file for this input)
    parity_checked = verify_outputs(args, rank, w, h, c, out_sets[(args.steps - 1) % n_sets], [r[0] for r in res])
    pixels_per_step = B * w * h
    value = world * pixels_per_step * args.steps / elapsed / 1e6
    run_values = [round(world * pixels_per_step * args.steps / r / 1e6, 1) for r in runs]

    # ---- per-kernel durations with HIP events on the encoder's own stream (untimed extra steps) ----
    enc.set_profiling(True)
    phases = np.zeros(8)
    reps = 5
    for _ in range(reps):
        step()
        phases += np.array(enc.last_phase_ms())
    phases /= reps
    enc.set_profiling(False)
    names = enc.phase_names()
    phase_ms = {n: round(float(phases[i]), 4) for i, n in enumerate(names)}
    alg_bytes = B * w * h * c + png_bytes  # input read once + PNG written once
    dom = max((k for k in names if k in ("encode_rows", "assemble", "hist")), key=lambda k: phase_ms[k])
    dom_s = phase_ms[dom] / 1e3
    achieved = alg_bytes / dom_s / 1e9
    kernels_s = sum(phase_ms[k] for k in names) / 1e3
    traffic, traffic_src = committed_traffic(f"{dom}_kernel", workload_tag(args))  # PMC passes of THIS command, if committed
    roofline = {"bound": "hbm", "kernel": f"{dom}_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": phase_ms[dom],
                "all_kernels_ms": round(kernels_s * 1e3, 4),
                "pipeline_frac": round(alg_bytes / kernels_s / 1e9 / HBM_PEAK_GBS, 4),
                "step_frac": round(alg_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),  # the timed, pipelined step against the same peak
                "phase_ms": phase_ms}
    # what the whole chain moves per step (all kernels, PMC) and the rate it moves it at in the timed, pipelined regions: the
    # number to hold against what plain streaming kernels reach on the same read/write mix (profiles/r03_mix_probe.txt: 5.1 TB/s)
    chain = committed_chain_traffic(workload_tag(args))
    if chain:
        roofline["chain_traffic"] = chain
        roofline["chain_traffic_rate_GBs"] = round(chain / (elapsed / args.steps) / 1e9, 1)

    line = {
        "metric": f"encode megapixels/sec ({scope_name(world)}), {PASS_NAME.get(args.flags, 'flags=%d' % args.flags)}, device-resident",
        "value": round(value, 1), "unit": "MP/s", "value_MiPs": round(value * 1e6 / 2 ** 20, 1), "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "prewarm": args.prewarm, "prewarm_bursts": 1 if (args.prewarm + args.warmup) else 0,  # (+ one untimed burst of `steps` submissions shaped like a timed region)
        "parity_checked": bool(parity_checked) if parity_checked is not None else None,
        "parity_images": parity_checked, "runs": run_values,
        "spread": round((max(run_values) - min(run_values)) / value, 4) if len(run_values) > 1 else None,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "per_rank_ms_per_step": per_rank_ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{B} x {w}x{h} {'RGBA' if c == 4 else 'RGB'} '{args.kind}' frames per GPU per step, "
                               f"flags={args.flags}, bit-exact fpng PNG output", "batch_per_gpu": B,
                   "width": w, "height": h, "channels": c, "png_bytes_per_step_per_gpu": png_bytes,
                   "parallelism": f"images sharded over {world} GPU(s), no data-path collective",
                   # how the submissions' chains share the GPU (fpng_amd_runtime_info(): the library asks the HIP runtime for eight hardware
                   # queues when it is loaded before the process's first HIP call; a process that comes too late runs two lanes over four)
                   "lanes": enc.lanes, "hw_queues": rt["hw_queues"], "hw_queue_source": rt["hw_queue_source"]},
        "roofline": roofline,
    }
    # ---- the way back: the files of the last submission, still in device memory, decoded to pixels in device memory ----
    def all_max(el):
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    last_set = out_sets[(args.steps - 1) % n_sets]
    pngs = [o[: r[0]] for o, r in zip(last_set, res)]
    dec = decode_bench(args, enc, imgs, pngs, w, h, c, rank, world, barrier, all_max, rank == 0 and world == 1 and not args.no_cpu_baseline)
    if True:
        if args.mode == "decode":  # the line is the decoder's; what the encoder did in this run goes along
            enc_part = {k: line[k] for k in ("metric", "value", "unit", "ms_per_step", "parity_checked", "parity_images", "roofline")}
            line = dict(line, **{k: dec[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "runs", "parity_checked", "parity_images", "roofline")})
            line["spread"] = round((max(dec["runs"]) - min(dec["runs"])) / dec["value"], 4) if len(dec["runs"]) > 1 else None
            line["warmup"], line["prewarm"] = 3, 0
            line["config"]["workload"] = (f"{B} fpng files ({w}x{h} {'RGBA' if c == 4 else 'RGB'} '{args.kind}', flags={args.flags}, as the encoder wrote them, "
                                          "device-resident) per GPU per step -> pixels in device memory, equal to the encoded frames")
            for k in ("cpu_baseline", "speedup_vs_cpu_1core"):
                if k in dec:
                    line[k] = dec[k]
            line["encode"] = enc_part
        else:
            line["decode"] = dec
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode != "decode":
        line["end_to_end"] = end_to_end(local_rank, w, h, c, args.kind, args.flags)
        line["cpu_baseline"] = cpu_baseline(w, h, c, args.kind, args.flags, args.cpu_reps)
        line["speedup_vs_cpu_1core"] = round(value / line["cpu_baseline"]["value"], 1)
    if rank == 0:
        errs = check_rates(line)
        if errs:
            line["reporting_errors"] = errs
            print("bench.py: " + "; ".join(errs), file=sys.stderr)
        _RESULT.append(json.dumps(line))
    enc.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


_RESULT = []  # the one JSON line, printed LAST: RCCL writes its version banner to stdout through C stdio whenever it likes


if __name__ == "__main__":
    main()
    if _RESULT:
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(_RESULT[-1], flush=True)
