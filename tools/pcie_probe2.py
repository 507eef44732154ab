#!/usr/bin/env python
"""Raw-HIP version of the streamed host path's copy pattern (no kernels): uploader thread = 8 chunks of 16.6 MB, each
hipMemcpyAsync + hipStreamSynchronize on its own non-blocking stream; downloader thread = chunk k of 7.25 MB after upload k
(+ an optional delay), on another stream.  Variants: stream flags, pinned or pageable host memory."""
import ctypes as C, os, sys, threading, time
import numpy as np
import torch  # (one HIP runtime in the process)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
H2D, D2H = 1, 2
IN, OUT, NB = 132710400, 58040724, 8
d_in, d_out = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(d_in), IN) == 0 and hip.hipMalloc(C.byref(d_out), OUT + 4096) == 0
h_in = np.ones(IN, dtype=np.uint8); h_out = np.ones(OUT + 4096, dtype=np.uint8)

def mk_stream(flags):
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), flags) == 0
    return s

def run(s_up, s_down, lag, trace=False):
    done = [0]; cv = threading.Condition(); tl = {}
    t0 = time.perf_counter()
    def up():
        n = IN // NB
        for k in range(NB):
            a = time.perf_counter()
            hip.hipMemcpyAsync(C.c_void_p(d_in.value + k * n), C.c_void_p(h_in.ctypes.data + k * n), n, H2D, s_up); hip.hipStreamSynchronize(s_up)
            tl["u%d" % k] = (a - t0, time.perf_counter() - t0)
            with cv:
                done[0] = k + 1; cv.notify_all()
    def down():
        n = (OUT // NB) & ~15
        for k in range(NB):
            with cv:
                cv.wait_for(lambda: done[0] > min(k + lag, NB - 1))
            a = time.perf_counter()
            hip.hipMemcpyAsync(C.c_void_p(h_out.ctypes.data + k * n), C.c_void_p(d_out.value + k * n), n, D2H, s_down); hip.hipStreamSynchronize(s_down)
            tl["d%d" % k] = (a - t0, time.perf_counter() - t0)
    a, b = threading.Thread(target=up), threading.Thread(target=down)
    a.start(); b.start(); a.join(); b.join()
    el = time.perf_counter() - t0
    if trace:
        print("   " + " ".join(f"{k}:{v[0]*1e6:.0f}-{v[1]*1e6:.0f}" for k, v in sorted(tl.items())))
    return el

VARIANT = sys.argv[1] if len(sys.argv) > 1 else ""
if VARIANT:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import fpng_amd
    su, sd = mk_stream(1), mk_stream(1)
    run(su, sd, 0)
    print(f"[{VARIANT}] before anything: {min(run(su, sd, 0) for _ in range(5))*1e3:.2f} ms")
    enc = fpng_amd.Encoder(device=0, stream="own")
    run(su, sd, 0)
    print(f"[{VARIANT}] with an Encoder alive: {min(run(su, sd, 0) for _ in range(5))*1e3:.2f} ms")
    w, h, c = 7680, 4320, 4
    img = fpng_amd.synth_image("grad", w, h, c)
    out = np.empty(fpng_amd.max_encoded_size(w, h, c), dtype=np.uint8)
    enc.encode_host_into(img, w, h, c, out, 0)
    run(su, sd, 0)
    print(f"[{VARIANT}] after one encode_host call: {min(run(su, sd, 0) for _ in range(5))*1e3:.2f} ms")
    h_in = img.reshape(-1); h_out = out
    run(su, sd, 0)
    print(f"[{VARIANT}] the encoder's host buffers (synth image in, np.empty out): {min(run(su, sd, 0) for _ in range(5))*1e3:.2f} ms")
    su2, sd2 = mk_stream(1), mk_stream(1)
    run(su2, sd2, 0)
    print(f"[{VARIANT}] fresh streams: {min(run(su2, sd2, 0) for _ in range(5))*1e3:.2f} ms")
    run(su2, sd2, 0, trace=True)
    sys.exit(0)
for name, fu, fd in [("non-blocking streams", 1, 1), ("default-flag streams", 0, 0)]:
    su, sd = mk_stream(fu), mk_stream(fd)
    for lag in (0, 1):
        run(su, sd, lag)
        best = min(run(su, sd, lag) for _ in range(5))
        print(f"{name}, download k waits for upload k+{lag}: {best*1e3:.2f} ms")
    run(su, sd, 0, trace=True)
# the same with both host buffers page-locked
assert hip.hipHostRegister(C.c_void_p(h_in.ctypes.data), C.c_size_t(IN), 0) == 0 and hip.hipHostRegister(C.c_void_p(h_out.ctypes.data), C.c_size_t(OUT + 4096), 0) == 0
su, sd = mk_stream(1), mk_stream(1)
run(su, sd, 0)
print(f"page-locked host buffers: {min(run(su, sd, 0) for _ in range(5))*1e3:.2f} ms")
run(su, sd, 0, trace=True)
