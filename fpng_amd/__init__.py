"""fpng_amd -- MI355X-native (gfx950) implementation of the fpng fast-PNG encode hot path.

The product is libfpng_amd.so (hand-written HIP kernels behind a C ABI, include/fpng_amd.h) plus
the `fpng::` C++ drop-in; this package is the Python door to the same C ABI.
"""
import os as _os
import sys as _sys

from . import _lib
if _os.path.exists(_lib.LIB_PATH) and "fpng_amd.build" not in getattr(_sys, "orig_argv", []):  # (`python -m fpng_amd.build` is about to replace it)
    # The library is loaded NOW, not at the first call: loading it asks the HIP runtime for eight hardware queues
    # (GPU_MAX_HW_QUEUES, csrc/api.cpp runtime_defaults(); fpng_amd.runtime_info() reports what came of it), which only counts
    # before the process's first HIP call -- import fpng_amd before the first torch.cuda call.
    # A library that cannot be loaded or is older than this package (a stale build: the rebuild imports this package first) must
    # not keep the package from being imported: the first real call reports it (there is no fallback behind it).
    try:
        _lib.load()
    except (OSError, AttributeError) as _e:
        import warnings as _warnings
        _warnings.warn(f"fpng_amd: {_lib.LIB_PATH} could not be loaded at import ({_e}); rebuild it with `python -m fpng_amd.build --force`")
from .api import (FPNG_ADLER32_INIT, FPNG_CRC32_INIT, FPNG_ENCODE_SLOWER, FPNG_FORCE_UNCOMPRESSED, MODE_COMPRESSED,  # noqa: F401
                  MODE_STORED, Encoder, FpngAmdError, adler32_combine, crc32_combine, fpng_adler32,
                  fpng_cpu_supports_sse41, fpng_crc32, fpng_encode_image_to_file, fpng_encode_image_to_memory,
                  fpng_init, layout_1pass, max_encoded_size, synth_image, Node, plan_bands, band_window, idat_crc_from_bands,
                  png_head, png_tail, pin_host_memory, unpin_host_memory, release_cached_memory, runtime_info)
