// fpng.h -- drop-in replacement header for richgel999/fpng's `namespace fpng` on MI355X.
//
// Same functions, signatures, constants and enum values as the reference's src/fpng.h:17-111, so
// existing callers recompile unchanged and link against libfpng.so (fpng_amd/csrc/fpng_dropin.cpp)
// instead of fpng.cpp.  The encode path runs on the GPU through the C ABI in fpng_amd.h and produces
// byte-identical files; decoding (fpng_amd/csrc/fpng_decode.cpp) goes through the GPU decoder for images of 256K pixels and
// more and through a CPU decoder for smaller ones -- same pixels, same status codes as the reference's decoder either way.
//
// Differences a caller can observe:
//   * fpng_init() binds a HIP device instead of probing CPUID (still optional);
//   * fpng_cpu_supports_sse41() answers "is the accelerator usable";
//   * if no GPU is usable the encode functions return false (there is no silent CPU path); the decode functions then use
//     their CPU decoder for every image;
//   * fpng_decode_memory() leaves `out` with the reference's size() on every exit (empty after a container-level failure, width *
//     height * desired_channels after success or a failure inside the stream: src/fpng.cpp:3087-3136), but a vector reused from call
//     to call is not zero-filled again (the reference resizes it to 0 and back on every call);
//   * FPNG_DISABLE_DECODE_CRC32_CHECKS (src/fpng.cpp:50-53) is honoured when the libraries are built with it
//     (python -m fpng_amd.build --variant nocrc -> libfpng_nocrc.so + libfpng_amd_nocrc.so).
#pragma once

#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace fpng {

// ---- initialisation (reference src/fpng.h:17) ----
void fpng_init();

// ---- utilities (reference src/fpng.h:23-31) ----
bool fpng_cpu_supports_sse41();

const uint32_t FPNG_CRC32_INIT = 0;
uint32_t fpng_crc32(const void *pData, size_t size, uint32_t prev_crc32 = FPNG_CRC32_INIT);

const uint32_t FPNG_ADLER32_INIT = 1;
uint32_t fpng_adler32(const void *pData, size_t size, uint32_t adler = FPNG_ADLER32_INIT);

// ---- compression (reference src/fpng.h:34-52) ----
enum {
    FPNG_ENCODE_SLOWER = 1,      // 2-pass: per-image Huffman table (~6% smaller)
    FPNG_FORCE_UNCOMPRESSED = 2, // stored Deflate blocks only
};

// pImage: R first in memory, pitch = w * num_chans, num_chans 3 or 4.  out_buf is resized and
// completely overwritten.
bool fpng_encode_image_to_memory(const void *pImage, uint32_t w, uint32_t h, uint32_t num_chans,
                                 std::vector<uint8_t> &out_buf, uint32_t flags = 0);

#ifndef FPNG_NO_STDIO
bool fpng_encode_image_to_file(const char *pFilename, const void *pImage, uint32_t w, uint32_t h, uint32_t num_chans,
                               uint32_t flags = 0);
#endif

// ---- decompression (reference src/fpng.h:55-111), CPU ----
enum {
    FPNG_DECODE_SUCCESS = 0,
    FPNG_DECODE_NOT_FPNG,
    FPNG_DECODE_INVALID_ARG,
    FPNG_DECODE_FAILED_NOT_PNG,
    FPNG_DECODE_FAILED_HEADER_CRC32,
    FPNG_DECODE_FAILED_INVALID_DIMENSIONS,
    FPNG_DECODE_FAILED_DIMENSIONS_TOO_LARGE,
    FPNG_DECODE_FAILED_CHUNK_PARSING,
    FPNG_DECODE_FAILED_INVALID_IDAT,
    FPNG_DECODE_FILE_OPEN_FAILED,
    FPNG_DECODE_FILE_TOO_LARGE,
    FPNG_DECODE_FILE_READ_FAILED,
    FPNG_DECODE_FILE_SEEK_FAILED
};

int fpng_get_info(const void *pImage, uint32_t image_size, uint32_t &width, uint32_t &height, uint32_t &channels_in_file);

int fpng_decode_memory(const void *pImage, uint32_t image_size, std::vector<uint8_t> &out, uint32_t &width, uint32_t &height,
                       uint32_t &channels_in_file, uint32_t desired_channels);

#ifndef FPNG_NO_STDIO
int fpng_decode_file(const char *pFilename, std::vector<uint8_t> &out, uint32_t &width, uint32_t &height,
                     uint32_t &channels_in_file, uint32_t desired_channels);
#endif

} // namespace fpng
