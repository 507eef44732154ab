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
VARIANTS = {"nocrc": ["FPNG_DISABLE_DECODE_CRC32_CHECKS=1"],  # the reference's fuzzing switch (src/fpng.cpp:50-53): libfpng_amd_nocrc.so + libfpng_nocrc.so
            "direct_defer": ["FPNG_DIRECT_SPIN_LIMIT=0"],  # every chunk that has to wait at all is deferred to scan_kernel (tests)
            "abl_nolook": ["FPNG_DIRECT_ABL=1"], "abl_nostore": ["FPNG_DIRECT_ABL=2"], "abl_nobehind": ["FPNG_DIRECT_ABL=4"], "abl_all": ["FPNG_DIRECT_ABL=7"],  # timing only: wrong files
            "direct_sleep4": ["FPNG_DIRECT_SLEEP=4"], "direct_sleep64": ["FPNG_DIRECT_SLEEP=64"],
            "dec_noprefilter": ["FPNG_DEC_PREFILTER=0"],
            "dec_pad27k": ["FPNG_DEC_PAD_LDS=27648"],  # decoder occupancy probe: two workgroups per compute unit instead of three
            "direct_w8": ["FPNG_DIRECT_WPE=8"], "direct_w5_win1536": ["FPNG_DIRECT_WPE=5", "FPNG_STAGE_DWORDS=1536", "FPNG_ROWS_WPE=5"], "direct_win1280": ["FPNG_STAGE_DWORDS=1280", "FPNG_ROWS_WPE=6"],
            "rows_w7": ["FPNG_ROWS_WPE=7"],  # the 3-channel and the narrow 4-channel walk at seven waves per SIMD too
            "rows4_w6": ["FPNG_ROWS_WPE4=6"],
            "rows4_w5": ["FPNG_ROWS_WPE4=5"], "rows4_w7": ["FPNG_ROWS_WPE4=7"],
            "rows4_w8": ["FPNG_ROWS_WPE4=8"],  # the 4-channel row kernel at eight waves per SIMD on wide rows too (the product: six there)
            "win1536_w5": ["FPNG_STAGE_DWORDS=1536", "FPNG_ROWS_WPE=5"], "win2048_w4": ["FPNG_STAGE_DWORDS=2048", "FPNG_ROWS_WPE=4"], "win1024_w6": ["FPNG_ROWS_WPE=6"], "win1024_w4": ["FPNG_ROWS_WPE=4"],
            "lead96": ["FPNG_DEC_LEADIN=96"], "lead64": ["FPNG_DEC_LEADIN=64"], "timing": ["FPNG_BUILD_TIMING"], "nont": ["FPNG_LOCAL_NT=0"], "rows8": ["FPNG_ROW_WAVES=8"], "rows2": ["FPNG_ROW_WAVES=2"],
            "emit_nostore": ["FPNG_DEC_EMIT_NOSTORE"], "emit_nt": ["FPNG_DEC_EMIT_NT"], "unf16": ["FPNG_DEC_UNF_ROWS=16"], "unf64": ["FPNG_DEC_UNF_ROWS=64"], "unf32": ["FPNG_DEC_UNF_ROWS=32"], "unf_w4": ["FPNG_DEC_UNF_WAVES=4"], "unf_w5": ["FPNG_DEC_UNF_WAVES=5"], "unf64_w4": ["FPNG_DEC_UNF_ROWS=64", "FPNG_DEC_UNF_WAVES=4"], "unf64_w5": ["FPNG_DEC_UNF_ROWS=64", "FPNG_DEC_UNF_WAVES=5"], "emit_l2store": ["FPNG_DEC_EMIT_L2STORE"],
            "dec_np": ["FPNG_DEC_PERSISTENT=0"], "dec_np6": ["FPNG_DEC_PERSISTENT=0", "FPNG_DEC_WGS=6"], "dec_npnv": ["FPNG_DEC_PERSISTENT=0", "FPNG_DEC_VOTE=0"],
            "dec_p6": ["FPNG_DEC_WGS=6"], "dec_nv": ["FPNG_DEC_VOTE=0"],
            "refix1": ["FPNG_DEC_REFIX_ROUNDS=1"], "refix2": ["FPNG_DEC_REFIX_ROUNDS=2"], "refix3": ["FPNG_DEC_REFIX_ROUNDS=3"], "refix5": ["FPNG_DEC_REFIX_ROUNDS=5"]}

if __name__ == "__main__":
    if "--variant" in sys.argv:
        v = sys.argv[sys.argv.index("--variant") + 1]
        print(build_variant(v, VARIANTS[v], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
