#!/usr/bin/env python
"""Turn a gpurun_out/prof_<tag>/ directory (tools/gpu_profile_round.sh) into the committed summaries:
profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and profiles/<tag>_pmc_traffic.txt."""
import collections, csv, json, shutil, sys, os

tag = sys.argv[1]
alg = int(sys.argv[2]) if len(sys.argv) > 2 else None
cal_tag = sys.argv[3] if len(sys.argv) > 3 else tag  # the run that holds the calibration streams (one per session is enough)
base = f"gpurun_out/prof_{tag}"
cal_base = f"gpurun_out/prof_{cal_tag}"
shutil.copy(f"{base}/trace/trace_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")


def load_all(path):
    """kernel name -> the counter's value of EVERY dispatch (templates of one kernel under one name)"""
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("fpng_amd::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "calib" not in name:
            name = name.split("<")[0]  # encode_rows_kernel<4> -> encode_rows_kernel
        agg[name].append(float(r["Counter_Value"]))
    return agg


def load(path):
    return {k: sum(v) / len(v) for k, v in load_all(path).items()}


def per_step(all_values, kernels):
    """launches per STEP of every kernel: a step's kernels are not launched equally often (dec_sync_kernel runs three times per
    decode call), so a sum of per-name averages undercounts (round 5's decode total did: 6.14 instead of ~6.5 GB).  The steps of the
    profiled run = the dispatches of a kernel that runs exactly once per step (scan_kernel / dec_offsets_kernel)."""
    anchor = next((a for a in ("dec_offsets_kernel", "scan_kernel") if a in kernels and a in all_values), None)
    counts = {k: len(all_values.get(k, ())) for k in kernels}
    steps = counts[anchor] if anchor else sorted(counts.values())[len(counts) // 2]
    return {k: counts[k] / max(steps, 1) for k in kernels}, steps, anchor


cal_f, cal_w = load(f"{cal_base}/cal_FETCH_SIZE/cal_counter_collection.csv"), load(f"{cal_base}/cal_WRITE_SIZE/cal_counter_collection.csv")
f_all, w_all = load_all(f"{base}/pmc_FETCH_SIZE/pmc_counter_collection.csv"), load_all(f"{base}/pmc_WRITE_SIZE/pmc_counter_collection.csv")
f, w = {k: sum(v) / len(v) for k, v in f_all.items()}, {k: sum(v) / len(v) for k, v in w_all.items()}
GiB = 1 << 30
fscale = GiB / [v for k, v in cal_f.items() if "calib_read_kernel<unsigned int>" in k][0]
wscale = GiB / [v for k, v in cal_w.items() if "calib_write_kernel<unsigned int>" in k][0]
args = " ".join(sys.argv[4:])
only = os.environ.get("PROFILE_KERNELS", "")  # e.g. dec_ : the decoder's kernels of a bench.py --mode decode run
# (the default bench.py run also decodes -- its `decode` object -- and checks pixels with torch: without PROFILE_KERNELS the summary
#  is the ENCODE chain's, the decoder's kernels have their own file)
skip = () if only else ("dec_", "at::")
lines = [f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs, --kernel-trace only) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline {args}".rstrip()
         + (f"  [kernels {only}*]" if only else ""),
         f"# calibration (tools/pmc_calibrate.py, 1 GiB streams with 4-byte lanes): FETCH_SIZE unit = {fscale:.1f} B, WRITE_SIZE unit = {wscale:.1f} B",
         "# per kernel: counter and megabytes of ONE launch (mean over its launches), its launches per step, megabytes per step",
         f"{'kernel':<22} {'FETCH_SIZE':>12} {'fetch_MB':>10} {'WRITE_SIZE':>12} {'write_MB':>10} {'per_step':>8} {'step_MB':>10}"]
tot = 0
kernels = [k for k in f if k.endswith("_kernel") and "calib" not in k and k.startswith(only) and not k.startswith(skip)]
launches, steps, anchor = per_step(f_all, kernels)
for k in sorted(kernels, key=lambda k: -launches[k] * (f.get(k, 0) * fscale + w.get(k, 0) * wscale)):
    fb, wb = f.get(k, 0) * fscale, w.get(k, 0) * wscale
    tot += launches[k] * (fb + wb)
    lines.append(f"{k:<22} {f.get(k,0):>12.1f} {fb/1e6:>10.1f} {w.get(k,0):>12.1f} {wb/1e6:>10.1f} {launches[k]:>8.2f} {launches[k]*(fb+wb)/1e6:>10.1f}")
lines.append(f"# steps in the profiled run: {steps} (launches of {anchor or 'the median kernel'})")
lines.append(f"total HBM-side traffic per launch: {tot/1e6:.1f} MB" + (f"  ({tot/alg:.2f} x the {alg} algorithmic bytes)" if alg else "")
             + "  [per STEP: every kernel weighted by its launches per step]")
open(f"profiles/{tag}_pmc_traffic.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")):
    print(r["Name"].replace("fpng_amd::(anonymous namespace)::", "").split("(")[0][:28], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
