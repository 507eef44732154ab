"""ctypes door to tools/probes/stream_probe.hip (a plain streaming reader / writer for PMC calibration and memory-system probes).
Built in place on first use (hipcc cross-compiles: build it in the container, the .so travels to the GPU box)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libstream_probe.so")
SRC = os.path.join(HERE, "stream_probe.hip")
_lib = None


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC])
    return LIB


def stream(buf, write, lane_bytes, stream=None):
    """Read (write=0) or write (write=1) the torch uint8 CUDA tensor `buf` once with 4- or 16-byte lanes, on torch's current stream."""
    global _lib
    import torch
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.stream_probe.restype = C.c_int
        _lib.stream_probe.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_size_t]
    s = torch.cuda.current_stream().cuda_stream if stream is None else stream
    rc = _lib.stream_probe(C.c_void_p(s) if s else None, int(write), lane_bytes, buf.data_ptr(), buf.numel())
    if rc:
        raise RuntimeError(f"stream_probe: {rc}")


if __name__ == "__main__":
    print(build())
