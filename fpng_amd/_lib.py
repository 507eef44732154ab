"""ctypes loader for libfpng_amd.so (the C ABI in include/fpng_amd.h).

torch is imported first ON PURPOSE: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7; loading
it first makes our library (DT_NEEDED libamdhip64.so.7) bind to the same HIP runtime instance
instead of pulling in a second one from /opt/rocm.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# FPNG_AMD_LIB: developer override for A/B-timing two builds on the same GPU box
LIB_PATH = os.environ.get("FPNG_AMD_LIB") or os.path.join(_HERE, "lib", "libfpng_amd.so")

NUM_PHASES = 8


class Image(C.Structure):
    _fields_ = [("d_pixels", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("num_chans", C.c_uint32),
                ("d_out", C.c_void_p), ("out_cap", C.c_size_t)]


class Result(C.Structure):
    _fields_ = [("png_size", C.c_uint64), ("mode", C.c_uint32), ("status", C.c_uint32)]


class HostImage(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("num_chans", C.c_uint32), ("reserved", C.c_uint32),
                ("out", C.c_void_p), ("out_cap", C.c_size_t), ("out_size", C.POINTER(C.c_size_t)), ("path", C.c_char_p)]


class Band(C.Structure):
    _fields_ = [("d_rows", C.c_void_p), ("d_row_above", C.c_void_p), ("w", C.c_uint32), ("num_chans", C.c_uint32),
                ("y0", C.c_uint32), ("y1", C.c_uint32), ("h_total", C.c_uint32), ("reserved", C.c_uint32)]


class BandStats(C.Structure):
    _fields_ = [("token_bits", C.c_uint64), ("adler_s1", C.c_uint32), ("adler_s2", C.c_uint32),
                ("adler_len", C.c_uint64), ("last_unit_bits", C.c_uint32), ("first_token_bit", C.c_uint32),
                ("eob_bits", C.c_uint32), ("reserved", C.c_uint32)]


class PngIn(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_uint32), ("reserved", C.c_uint32), ("d_pixels", C.c_void_p), ("pixels_cap", C.c_size_t)]


class DecodeResult(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("channels_in_file", C.c_uint32), ("status", C.c_int32)]


class BandPlan(C.Structure):
    _fields_ = [("end_bit", C.c_uint64), ("zlib_size", C.c_uint64), ("adler", C.c_uint32), ("stored", C.c_uint32)]


RESERVE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class Transport(C.Structure):
    """fpng_amd_transport: the exchange functions fpng_amd_encode_image_sharded() calls (all on the encoder's stream)."""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("world", C.c_int),
                ("all_gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)),
                ("all_reduce_sum_u32", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)),
                ("group_begin", C.CFUNCTYPE(C.c_int, C.c_void_p)),
                ("send", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)),
                ("recv", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)),
                ("group_end", C.CFUNCTYPE(C.c_int, C.c_void_p))]

# every symbol include/fpng_amd.h declares: (restype, argtypes)
_u32, _u64, _sz, _vp, _int = C.c_uint32, C.c_uint64, C.c_size_t, C.c_void_p, C.c_int


class RuntimeInfo(C.Structure):
    """fpng_amd_runtime (include/fpng_amd.h)"""
    _fields_ = [("hw_queues", _u32), ("hw_queue_source", _u32), ("lanes", _u32), ("reserved", _u32 * 5)]


SIGNATURES = {
    "fpng_amd_init": (_int, [_int]),
    "fpng_amd_device_available": (_int, []),
    "fpng_amd_device_count": (_int, []),
    "fpng_amd_last_error": (C.c_char_p, []),
    "fpng_amd_abi_version": (_int, []),
    "fpng_amd_crc32": (_u32, [_vp, _sz, _u32]),
    "fpng_amd_adler32": (_u32, [_vp, _sz, _u32]),
    "fpng_amd_crc32_combine": (_u32, [_u32, _u32, _u64]),
    "fpng_amd_adler32_combine": (_u32, [_u32, _u32, _u64]),
    "fpng_amd_max_encoded_size": (_sz, [_u32, _u32, _u32]),
    "fpng_amd_encoder_create": (_int, [C.POINTER(_vp), _int, _vp]),
    "fpng_amd_encoder_create_on_stream": (_int, [C.POINTER(_vp), _int, _vp]),
    "fpng_amd_encoder_set_stream": (_int, [_vp, _vp]),
    "fpng_amd_encoder_destroy": (None, [_vp]),
    "fpng_amd_encoder_stream": (_vp, [_vp]),
    "fpng_amd_encoder_join": (_int, [_vp]),
    "fpng_amd_encoder_phase_names": (C.c_char_p, [_vp]),
    "fpng_amd_encode_batch_async": (_int, [_vp, C.POINTER(Image), _u32, _u32]),
    "fpng_amd_encode_finish": (_int, [_vp, C.POINTER(Result), _u32]),
    "fpng_amd_encode_submit": (_int, [_vp, C.POINTER(Image), _u32, _u32, C.POINTER(_u64)]),
    "fpng_amd_encode_wait": (_int, [_vp, _u64, C.POINTER(Result), _u32]),
    "fpng_amd_encode_query": (_int, [_vp, _u64]),
    "fpng_amd_encode_host": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _sz, C.POINTER(_sz)]),
    "fpng_amd_encode_host_batch": (_int, [_vp, C.POINTER(HostImage), _u32, _u32, _int]),
    "fpng_amd_band_hist": (_int, [_vp, C.POINTER(Band), _vp]),
    "fpng_amd_band_encode": (_int, [_vp, C.POINTER(Band), _u32, _vp, C.POINTER(BandStats)]),
    "fpng_amd_band_place": (_int, [_vp, C.POINTER(Band), _u64, _u64, _vp, _sz, C.POINTER(_u64), C.POINTER(_sz)]),
    "fpng_amd_1pass_layout": (_int, [_u32, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "fpng_amd_wrap_png": (_int, [_vp, _vp, _sz, _u32, _u32, _u32, _u32, C.POINTER(_sz)]),
    "fpng_amd_wrap_png_crc": (_int, [_vp, _vp, _sz, _u32, _u32, _u32, _u32, _vp, _u32, C.POINTER(_sz)]),
    "fpng_amd_band_crc_partials": (_int, [_vp, _vp, _u32, C.POINTER(_u32)]),
    "fpng_amd_band_crc": (_int, [_vp, C.POINTER(_u32), C.POINTER(_u64)]),
    "fpng_amd_plan_bands": (_int, [C.POINTER(BandStats), _u32, _u32, _u32, _u32, _u32, C.POINTER(_u64), C.POINTER(BandPlan)]),
    "fpng_amd_band_window": (_int, [_int, _int, _u64, _u64, _u32, C.POINTER(_u64), C.POINTER(_sz), C.POINTER(_u32)]),
    "fpng_amd_idat_crc_from_bands": (_u32, [C.POINTER(_u32), C.POINTER(_u64), _u32, _u64, _u32]),
    "fpng_amd_png_head": (_int, [_u32, _u32, _u32, _u64, _vp]),
    "fpng_amd_png_tail": (None, [_u32, _u32, _vp]),
    "fpng_amd_encode_host_to": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, RESERVE_FN, _vp, C.POINTER(_sz)]),
    "fpng_amd_rccl_unique_id": (_int, [_vp]),
    "fpng_amd_rccl_transport_create": (_int, [C.POINTER(C.POINTER(Transport)), _vp, _int, _int, _int]),
    "fpng_amd_rccl_transport_destroy": (None, [C.POINTER(Transport)]),
    "fpng_amd_encode_image_sharded": (_int, [_vp, C.POINTER(Transport), C.POINTER(Band), _u32, _int, _vp, _sz, C.POINTER(_sz)]),
    "fpng_amd_sharded_last_report": (_int, [_vp, _vp]),
    "fpng_amd_encoder_last_host_bands": (_int, [_vp]),
    "fpng_amd_pin_host_memory": (_int, [_vp, _sz]),
    "fpng_amd_unpin_host_memory": (_int, [_vp]),
    "fpng_amd_node_create": (_int, [C.POINTER(_vp), C.POINTER(_int), _u32]),
    "fpng_amd_node_destroy": (None, [_vp]),
    "fpng_amd_node_size": (_u32, [_vp]),
    "fpng_amd_node_encode_host_batch": (_int, [_vp, C.POINTER(HostImage), _u32, _u32, _int]),
    "fpng_amd_node_encode_host_image": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, RESERVE_FN, _vp, C.POINTER(_sz)]),
    "fpng_amd_decode_batch": (_int, [_vp, C.POINTER(PngIn), _u32, _u32, C.POINTER(DecodeResult)]),
    "fpng_amd_decode_batch_device": (_int, [_vp, C.POINTER(PngIn), _u32, _u32, C.POINTER(DecodeResult)]),
    "fpng_amd_decode_last_phase_ms": (_int, [_vp, C.POINTER(C.c_float * 4)]),
    "fpng_amd_decode_host": (_int, [_vp, _vp, _u32, _u32, RESERVE_FN, _vp, C.POINTER(DecodeResult)]),
    "fpng_amd_decode_plan": (_int, [_vp, _u32, C.POINTER(DecodeResult), C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32), C.POINTER(C.c_uint64),
                             C.POINTER(C.c_uint64), C.POINTER(_u32 * 4160)]),
    "fpng_amd_train_tables": (_int, [_vp, C.POINTER(Image), _u32, _u32, _vp, _sz, C.POINTER(_sz), C.POINTER(_u32), C.POINTER(_u32), _vp, _vp]),
    "fpng_amd_synth_image": (_int, [_int, _u32, _u32, _u32, _u32, _vp]),
    "fpng_amd_encoder_set_profiling": (_int, [_vp, _int]),
    "fpng_amd_encoder_last_phase_ms": (_int, [_vp, C.POINTER(C.c_float * NUM_PHASES)]),
    "fpng_amd_debug_peek": (_int, [_vp, _int, C.POINTER(_u32), _u32]),
    "fpng_amd_runtime_info": (_int, [C.POINTER(RuntimeInfo)]),
    "fpng_amd_encoder_lanes": (_u32, [_vp]),
    "fpng_amd_release_cached_memory": (_int, []),
}

_lib = None


def load():
    """Load the library (built by `python -m fpng_amd.build` / __graft_entry__.build()).
    Fails loudly if it is missing: there is no Python or CPU fallback for the encode path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -m fpng_amd.build` "
                          "(fpng_amd has no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class FpngAmdError(RuntimeError):
    def __init__(self, code, what):
        super().__init__(f"fpng_amd error {code}: {what}")
        self.code = code


def check(rc):
    if rc != 0:
        raise FpngAmdError(rc, load().fpng_amd_last_error().decode("utf-8", "replace"))
