// fpng_dropin.cpp -- `namespace fpng` (include/fpng.h) on top of the C ABI of libfpng_amd.so.
//
// Built by g++ into libfpng.so; it contains no HIP code and no encoder of its own: every encode call
// goes through fpng_amd_encode_host_to() (H2D copies, HIP kernels and D2H copies, streamed in row bands for large frames).  Each thread gets its own
// encoder object, so the functions stay re-entrant like the reference's.
//
// Decoding (fpng_get_info / fpng_decode_memory / fpng_decode_file): fpng_decode.cpp -- images of 256K pixels and more go
// through the GPU decoder (fpng_amd_decode_host), small ones and the files it leaves undecided through the CPU decoder there.
#include "fpng.h"

#include "fpng_amd.h"

#include <stdio.h>

namespace fpng {

namespace {
struct ThreadEncoder {
    fpng_amd_encoder *enc = nullptr;
    bool tried = false; // (a process without a GPU asks once per thread, not once per call)
    ~ThreadEncoder()
    {
        if (enc) fpng_amd_encoder_destroy(enc);
    }
    fpng_amd_encoder *get()
    {
        if (!enc && !tried) {
            tried = true;
            if (fpng_amd_encoder_create(&enc, -1, nullptr) != FPNG_AMD_OK) enc = nullptr;
        }
        return enc;
    }
};
thread_local ThreadEncoder t_encoder;
} // namespace

// the calling thread's encoder object (also used by the GPU tier of fpng_decode_memory, fpng_decode.cpp); NULL without a GPU
fpng_amd_encoder *dropin_thread_encoder() { return t_encoder.get(); }

void fpng_init() { (void)fpng_amd_init(-1); }

bool fpng_cpu_supports_sse41() { return fpng_amd_device_available() != 0; }

uint32_t fpng_crc32(const void *pData, size_t size, uint32_t prev_crc32) { return fpng_amd_crc32(pData, size, prev_crc32); }

uint32_t fpng_adler32(const void *pData, size_t size, uint32_t adler) { return fpng_amd_adler32(pData, size, adler); }

// reference src/fpng.cpp:1662-1803: returns false on bad arguments; "does not compress" is not an
// error (stored-block fallback happens on the device exactly where the reference would fall back).
bool fpng_encode_image_to_memory(const void *pImage, uint32_t w, uint32_t h, uint32_t num_chans, std::vector<uint8_t> &out_buf,
                                 uint32_t flags)
{
    if (!pImage) return false;
    if ((w < 1) || (h < 1) || ((uint64_t)w * h > UINT32_MAX) || (w > (1u << 24)) || (h > (1u << 24))) return false;
    if ((num_chans != 3) && (num_chans != 4)) return false;
    fpng_amd_encoder *enc = t_encoder.get();
    if (!enc) return false;
    // The vector is the output allocator: the encoder asks for room as the file's size becomes known (once, up front, for
    // an estimate; once at the end for the exact size), so nothing is sized for the worst case and zero-filled (the
    // reference resizes to the worst case, src/fpng.cpp:1686-1691, then shrinks) -- a vector reused from frame to frame, as
    // in the reference's own timing loop (fpng_test.cpp:1198-1209), is not touched at all until the bytes arrive.
    size_t size = 0;
    if (fpng_amd_encode_host_to(enc, pImage, w, h, num_chans, flags,
                                [](void *user, size_t bytes) -> uint8_t * {
                                    auto *v = static_cast<std::vector<uint8_t> *>(user);
                                    if (v->size() < bytes) v->resize(bytes);
                                    return v->data();
                                },
                                &out_buf, &size) != FPNG_AMD_OK) {
        out_buf.resize(0);
        return false;
    }
    out_buf.resize(size);
    return true;
}

#ifndef FPNG_NO_STDIO
// reference src/fpng.cpp:1806-1828
bool fpng_encode_image_to_file(const char *pFilename, const void *pImage, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t flags)
{
    std::vector<uint8_t> out_buf;
    if (!fpng_encode_image_to_memory(pImage, w, h, num_chans, out_buf, flags)) return false;
    FILE *f = fopen(pFilename, "wb");
    if (!f) return false;
    if (fwrite(out_buf.data(), 1, out_buf.size(), f) != out_buf.size()) {
        fclose(f);
        return false;
    }
    return fclose(f) != EOF;
}
#endif

} // namespace fpng
