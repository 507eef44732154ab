#!/usr/bin/env python
"""gpurun_out/sq_<tag>/ (tools/gpu_sq_counters.sh: two rocprofv3 --pmc passes of SQ counters over `bench.py --steps 2`)
-> profiles/<tag>_sq_counters.txt: per kernel the raw averages per launch and the per-64-pixel instruction counts."""
import collections, csv, sys

tag, pixels = sys.argv[1], int(sys.argv[2])


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("fpng_amd::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


a, b = load(f"gpurun_out/sq_{tag}/sq_counter_collection.csv"), load(f"gpurun_out/sq_{tag}/b/sq_counter_collection.csv")
out = [f"# rocprofv3 --pmc <8 SQ counters> --kernel-trace (two passes) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline   [{pixels} pixels per launch]",
       "# per launch averages; *_per64px = count / (pixels / 64).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* are quad-cycles summed over waves:",
       "# WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES."]
for k in sorted(a, key=lambda k: -a[k].get("SQ_WAVE_CYCLES", 0)):
    if "kernel" not in k or "rocclr" in k:
        continue
    d = dict(a[k])
    d.update(b.get(k, {}))
    w = d.get("SQ_WAVE_CYCLES", 1) or 1
    out.append(f"{k}")
    out.append("   per 64 px: VALU %.2f  SALU %.2f  LDS %.2f  VMEM_RD %.3f  VMEM_WR %.3f  SMEM %.3f" % tuple(
        d.get(c, 0) / (pixels / 64) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")))
    out.append("   wave time: parked %.1f %%  issue-stalled %.1f %%  issuing %.1f %%  (of which VALU %.1f %%, LDS %.1f %%); LDS bank-conflict cycles / LDS active cycles %.2f; waves %d" % (
        100 * d.get("SQ_WAIT_ANY", 0) / w, 100 * d.get("SQ_WAIT_INST_ANY", 0) / w, 100 * d.get("SQ_ACTIVE_INST_ANY", 0) / w,
        100 * d.get("SQ_ACTIVE_INST_VALU", 0) / w, 100 * d.get("SQ_ACTIVE_INST_LDS", 0) / w,
        d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, d.get("SQ_LDS_IDX_ACTIVE", 1)), d.get("SQ_WAVES", 0)))
    out.append("   raw: " + "  ".join(f"{c}={v:.0f}" for c, v in sorted(d.items())))
open(f"profiles/{tag}_sq_counters.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
