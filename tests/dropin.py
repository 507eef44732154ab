"""Loads the `namespace fpng` C++ drop-in (fpng_amd/lib/libfpng.so) through a tiny test-only C shim."""
import ctypes as C
import os
import subprocess

import numpy as np

import torch  # noqa: F401  keep one HIP runtime in the process (see fpng_amd/_lib.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_shim = None


def shim():
    global _shim
    if _shim is not None:
        return _shim
    from fpng_amd import build
    build.build()
    lib_dir = os.path.join(ROOT, "fpng_amd", "lib")
    src = os.path.join(ROOT, "tests", "cpp", "dropin_shim.cpp")
    so = os.path.join(lib_dir, "libfpng_test_shim.so")
    deps = [src, os.path.join(lib_dir, "libfpng.so")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), src, "-o", so,
                               "-L", lib_dir, "-lfpng", "-lfpng_amd", "-pthread", "-Wl,-rpath,$ORIGIN"])
    L = C.CDLL(so)
    L.shim_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.shim_encode_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.shim_get_info.argtypes = [C.c_void_p, C.c_uint32] + [C.POINTER(C.c_uint32)] * 3
    L.shim_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t] + [C.POINTER(C.c_uint32)] * 3 + [C.c_uint32]
    L.shim_decode_file.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t] + [C.POINTER(C.c_uint32)] * 3 + [C.c_uint32]
    L.shim_encode_threads.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.POINTER(C.c_int)]
    L.shim_decode_threads.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.shim_time_encode.restype = C.c_double
    L.shim_time_decode.restype = C.c_double
    L.shim_time_decode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
    L.shim_gpu_decodes.restype = C.c_ulonglong
    L.shim_time_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    L.shim_crc32.restype = C.c_uint32
    L.shim_crc32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    L.shim_adler32.restype = C.c_uint32
    L.shim_adler32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    _shim = L
    return L


def decode(png, desired):
    L = shim()
    b = np.frombuffer(bytes(png), dtype=np.uint8)
    w, h, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    cap = max(1 << 16, 4 * len(png) * 64)
    st = L.shim_get_info(b.ctypes.data, b.size, C.byref(w), C.byref(h), C.byref(c))
    if st == 0:
        cap = w.value * h.value * desired + 16
    out = np.zeros(cap, dtype=np.uint8)
    st = L.shim_decode(b.ctypes.data, b.size, out.ctypes.data, cap, C.byref(w), C.byref(h), C.byref(c), desired)
    return st, (out[: w.value * h.value * desired] if st == 0 else None), w.value, h.value, c.value


def decode_file(path, desired):
    """fpng::fpng_decode_file(path) -> (status, pixels or None, w, h, c)"""
    L = shim()
    w, h, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    cap = max(1 << 16, os.path.getsize(path) * 300 + (1 << 20)) if os.path.exists(path) else 1 << 16
    cap = min(cap, 1 << 30)
    out = np.zeros(cap, dtype=np.uint8)
    st = L.shim_decode_file(path.encode(), out.ctypes.data, cap, C.byref(w), C.byref(h), C.byref(c), desired)
    return st, (out[: w.value * h.value * desired] if st == 0 else None), w.value, h.value, c.value


def get_info(png):
    L = shim()
    b = np.frombuffer(bytes(png), dtype=np.uint8)
    w, h, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    st = L.shim_get_info(b.ctypes.data, b.size, C.byref(w), C.byref(h), C.byref(c))
    return st, w.value, h.value, c.value


def encode(img, w, h, c, flags=0):
    L = shim()
    a = np.ascontiguousarray(img, dtype=np.uint8)
    n_f = (w * c + 1) * h
    cap = 58 + 6 + n_f + 5 * ((n_f + 65534) // 65535) + 16 + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    ok = L.shim_encode(a.ctypes.data, w, h, c, flags, out.ctypes.data, cap, C.byref(n))
    return out[: n.value].tobytes() if ok else None


def time_decode(png, desired, reps=4):
    """Best seconds per fpng::fpng_decode_memory() call through libfpng.so into one reused std::vector."""
    b = np.frombuffer(bytes(png), dtype=np.uint8)
    t = shim().shim_time_decode(b.ctypes.data, b.size, desired, reps)
    if t < 0:
        raise RuntimeError("fpng::fpng_decode_memory failed")
    return t


def gpu_decodes():
    """Calls of fpng::fpng_decode_memory in this process that were answered by the GPU tier."""
    return int(shim().shim_gpu_decodes())


def time_encode(img, w, h, c, flags=0, reps=5, reuse=True):
    """Best seconds per fpng::fpng_encode_image_to_memory() call through libfpng.so, timed in C++ like the reference's
    harness (fpng_test.cpp:1198-1209); reuse: one std::vector across the calls / a fresh one per call."""
    L = shim()
    a = np.ascontiguousarray(img, dtype=np.uint8)
    n = C.c_size_t(0)
    t = L.shim_time_encode(a.ctypes.data, w, h, c, flags, reps, int(reuse), C.byref(n))
    if t < 0:
        raise RuntimeError("fpng::fpng_encode_image_to_memory failed")
    return t, n.value


def decode_threads(pngs, desired, reps=3):
    """fpng::fpng_decode_memory() from len(pngs) threads at once, every thread its own file `reps` times:
    (list of (status, pixels), True when every repetition of every thread gave the same result)."""
    L = shim()
    n = len(pngs)
    bufs = [np.frombuffer(bytes(p), dtype=np.uint8) for p in pngs]
    cap = 16
    for p in pngs:
        st, w, h, c = get_info(p)
        if st == 0:
            cap = max(cap, w * h * desired)
    outs = [np.zeros(cap, dtype=np.uint8) for _ in range(n)]
    P = C.c_void_p * n
    status = (C.c_int * n)()
    out_sizes = (C.c_size_t * n)()
    agree = C.c_int(0)
    L.shim_decode_threads(n, P(*[b.ctypes.data for b in bufs]), (C.c_uint32 * n)(*[b.size for b in bufs]), desired, reps, P(*[o.ctypes.data for o in outs]), cap,
                          status, out_sizes, C.byref(agree))
    return [(int(status[t]), outs[t][: out_sizes[t]]) for t in range(n)], bool(agree.value)


def encode_threads(images, flags, reps=3):
    """fpng::fpng_encode_image_to_memory() from len(images) threads at once, every thread its own image `reps` times:
    (list of PNG bytes, True when every repetition of every thread gave the same bytes)."""
    L = shim()
    n = len(images)
    arrs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
    import fpng_amd
    cap = max(fpng_amd.max_encoded_size(a.shape[1], a.shape[0], a.shape[2]) for a in arrs)
    outs = [np.empty(cap, dtype=np.uint8) for _ in range(n)]
    P = C.c_void_p * n
    U = C.c_uint32 * n
    sizes = (C.c_size_t * n)()
    agree = C.c_int(0)
    ok = L.shim_encode_threads(n, P(*[a.ctypes.data for a in arrs]), U(*[a.shape[1] for a in arrs]), U(*[a.shape[0] for a in arrs]),
                               U(*[a.shape[2] for a in arrs]), U(*flags), reps, P(*[o.ctypes.data for o in outs]), cap, sizes, C.byref(agree))
    if not ok:
        raise RuntimeError("fpng::fpng_encode_image_to_memory failed in a thread")
    return [outs[i][: sizes[i]].tobytes() for i in range(n)], bool(agree.value)


_sharded = None


def sharded_local():
    """tests/cpp/sharded_local.cpp: fpng_amd_encode_image_sharded() driven by N threads of this process over an in-process
    transport (the multi-rank logic on a one-GPU box)."""
    global _sharded
    if _sharded is not None:
        return _sharded
    from fpng_amd import build
    build.build()
    lib_dir = os.path.join(ROOT, "fpng_amd", "lib")
    src = os.path.join(ROOT, "tests", "cpp", "sharded_local.cpp")
    so = os.path.join(lib_dir, "libfpng_test_sharded.so")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    deps = [src, os.path.join(lib_dir, "libfpng_amd.so"), os.path.join(ROOT, "include", "fpng_amd.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(rocm, "include"), src, "-o", so, "-L", lib_dir, "-lfpng_amd", "-L", os.path.join(rocm, "lib"),
                               "-lamdhip64", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    L = C.CDLL(so)
    L.shim_sharded_local.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                     C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    L.shim_sharded_local2.argtypes = L.shim_sharded_local.argtypes + [C.c_void_p]
    _sharded = L
    return L


class ShardedReport(C.Structure):  # include/fpng_amd.h: fpng_amd_sharded_report
    _fields_ = [("sent_bytes", C.c_uint64), ("received_in_place", C.c_uint64), ("root_staged_bytes", C.c_uint64), ("own_window_bytes", C.c_uint64),
                ("shared_pieces", C.c_uint32), ("collectives", C.c_uint32), ("stored", C.c_uint32), ("reserved", C.c_uint32)]


def encode_sharded_local(img, cuts, flags=0, root=0, reports=None):
    """img: uint8 (h, w, c); cuts: row boundaries [0, ..., h] (one band per rank) -> PNG bytes from rank `root`.
    reports: a list that receives every rank's data-movement report (ShardedReport)."""
    L = sharded_local()
    a = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = a.shape
    world = len(cuts) - 1
    n_f = (w * c + 1) * h
    cap = 58 + 6 + n_f + 5 * ((n_f + 65534) // 65535) + 16 + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    err = C.create_string_buffer(512)
    rep = (ShardedReport * world)()
    rc = L.shim_sharded_local2(world, (C.c_uint32 * (world + 1))(*cuts), a.ctypes.data, w, h, c, flags, root, out.ctypes.data, cap, C.byref(n), err, 512, C.byref(rep))
    if rc:
        raise RuntimeError(f"sharded_local rc={rc}: {err.value.decode(errors='replace')}")
    if reports is not None:
        reports.extend(rep)
    return out[: n.value].tobytes()
