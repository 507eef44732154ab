"""Build libfpng_amd.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m fpng_amd.build            # rebuild if sources are newer than the library
    python -m fpng_amd.build --force

hipcc cross-compiles without a GPU, so this also runs in the CPU-only dev container.  The .so is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libfpng_amd.so")
DROPIN_LIB = os.path.join(LIB_DIR, "libfpng.so")
SOURCES = ["kernels.hip", "decode.hip", "api.cpp", "pipeline.cpp", "sharded.cpp", "decode_api.cpp", "format.cpp", "synth.cpp"]
HEADERS = [os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "format.h"), os.path.join(CSRC, "encoder.h"), os.path.join(CSRC, "decode.h"), os.path.join(CSRC, "decode_core.h"), os.path.join(CSRC, "host_workers.h"), os.path.join(CSRC, "png_parse.h"),
           os.path.join(ROOT, "include", "fpng_amd.h")]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_variant(name, defines, verbose=False):
    """A diagnostic build next to the product (e.g. --variant timing: per-phase cycle counters in encode_image_kernel);
    select it at run time with FPNG_AMD_LIB=fpng_amd/lib/libfpng_amd_<name>.so."""
    os.makedirs(LIB_DIR, exist_ok=True)
    out = os.path.join(LIB_DIR, f"libfpng_amd_{name}.so")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wall",
           "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", out, "-ldl"] + [f"-D{d}" for d in defines]
    for s in SOURCES:
        cmd += ["-x", "hip", os.path.join(CSRC, s)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if name in DROPIN_VARIANTS:  # options of png_parse.h / fpng_decode.cpp: the `namespace fpng` library is built against the variant too
        dropin = os.path.join(LIB_DIR, f"libfpng_{name}.so")
        cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + [f"-D{d}" for d in defines] + [
            os.path.join(CSRC, "fpng_dropin.cpp"), os.path.join(CSRC, "fpng_decode.cpp"), "-o", dropin, "-L", LIB_DIR, f"-lfpng_amd_{name}", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if force or _stale(LIB, srcs + HEADERS):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
               "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB, "-ldl"]
        for s in srcs:
            cmd += ["-x", "hip", s]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    dropin_srcs = [os.path.join(CSRC, "fpng_dropin.cpp"), os.path.join(CSRC, "fpng_decode.cpp")]
    if force or _stale(DROPIN_LIB, dropin_srcs + [LIB, os.path.join(ROOT, "include", "fpng.h"), os.path.join(CSRC, "png_parse.h")]):
        # the `namespace fpng` drop-in: plain C++ over the C ABI, no HIP in it
        cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + dropin_srcs + [
            "-o", DROPIN_LIB, "-L", LIB_DIR, "-lfpng_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # the fpng_test-style command line harness (tools/fpng_amd_test.cpp) over the drop-in
    cli_src = os.path.join(ROOT, "tools", "fpng_amd_test.cpp")
    cli = os.path.join(LIB_DIR, "fpng_amd_test")
    if os.path.exists(cli_src) and (force or _stale(cli, [cli_src, os.path.join(ROOT, "tools", "png_loader.h"), LIB, DROPIN_LIB, os.path.join(ROOT, "include", "fpng.h")])):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(rocm, "include"),
               cli_src, "-o", cli, "-L", LIB_DIR, "-lfpng", "-lfpng_amd", "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-ldl", "-lpthread",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
        if verbose:
            print(" ".join(cmd))
        try:  # the harness is optional: both libraries are complete without it
            subprocess.check_call(cmd)
        except (subprocess.CalledProcessError, OSError) as e:
            print(f"fpng_amd.build: warning: could not link the command line harness ({e}); continuing", file=sys.stderr)
    return LIB


DROPIN_VARIANTS = {"nocrc"}
# Build variants next to the product: the reference's own build switch, and the diagnostic builds a committed profile was made with
# (the A/B variants of rounds 3-5 whose outcome is settled are gone from the sources; `git log -S<define>` finds them).
VARIANTS = {"nocrc": ["FPNG_DISABLE_DECODE_CRC32_CHECKS=1"],  # the reference's fuzzing switch (src/fpng.cpp:50-53): libfpng_amd_nocrc.so + libfpng_nocrc.so
            "timing": ["FPNG_BUILD_TIMING"],  # cycle counters inside build_dynamic_kernel (tools/build_timing.py, profiles/r05_table_builder.txt)
            "dec_pad27k": ["FPNG_DEC_PAD_LDS=27648"],  # decoder occupancy probe: two workgroups per compute unit instead of three (profiles/r05_decode_occupancy.txt)
            "sync_timing": ["FPNG_DEC_SYNC_TIMING"],  # dec_sync_kernel<false> stamps every workgroup: start / bits staged / first wave decoded / all decoded / corrected / end (FPNG_AMD_SYNC_TIMES=<file>; tools/gpu_sync_times.sh)
            "tile_timing": ["FPNG_DEC_TILE_TIMING"],  # dec_unfilter_kernel stamps every tile's start / rows there / carry known / end (FPNG_AMD_TILE_TIMES=<file>; profiles/r06y_tile_times.txt, r07b_tile_times.txt; tools/gpu_tile_times.sh)
            "rows4_w8": ["FPNG_ROWS_WPE4=8"]}  # the 4-channel row walk at eight waves per SIMD on wide rows too (profiles/r05_rows_w6.txt; the product: seven)

if __name__ == "__main__":
    if "--variant" in sys.argv:
        v = sys.argv[sys.argv.index("--variant") + 1]
        print(build_variant(v, VARIANTS[v], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
