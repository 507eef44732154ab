// fpng_decode.cpp -- CPU decoder half of the `namespace fpng` drop-in (include/fpng.h).
//
// Out of the GPU scope by design (a fpng stream is one serial Huffman bit string with
// no restart points), but the drop-in must keep the functions.  This is an independent restricted
// inflater that accepts exactly the files the reference's decoder accepts and reports the same
// status codes:
//   container walk ......... reference src/fpng.cpp:2930-3077 (fpng_get_info_internal)
//   dynamic-block header ... reference src/fpng.cpp:1954-2105 (prepare_dynamic_block), table
//                            completeness rule :1836-1862
//   pixel stream rules ..... reference src/fpng.cpp:2209-2584 / :2587-2901: row filter literal must
//                            be 0 then 2, a literal pixel is `chans` literals, a match is an RLE
//                            repeat of the previous pixel's DELTA (its 1-bit distance is skipped),
//                            matches are whole pixels and never cross a row, EOB + byte alignment
//                            must land exactly 4 bytes (the Adler-32, unchecked there too) before
//                            the end of the IDAT payload
//   stored blocks .......... reference src/fpng.cpp:2107-2207
// Any violation inside the zlib stream maps to FPNG_DECODE_NOT_FPNG (reference :3131-3136).
#include "png_parse.h"

#include "fpng_amd.h"

#include <atomic>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace fpng {

fpng_amd_encoder *dropin_thread_encoder(); // fpng_dropin.cpp

namespace {

constexpr uint64_t kGpuDecodeMinPixels = 1u << 18; // images of 512 x 512 pixels and more are decoded on the GPU
std::atomic<uint64_t> g_gpu_decodes{0};

using namespace parse;

bool inflate_stored(const uint8_t *z, uint32_t avail, uint32_t zlib_len, uint8_t *dst, uint32_t w, uint32_t h, uint32_t src_chans,
                    uint32_t dst_chans)
{
    const uint64_t src_bpl = (uint64_t)w * src_chans, dst_len = (uint64_t)w * dst_chans * h;
    uint64_t src = 2, out = 0, raster = 0;
    uint32_t comp = 0;
    for (;;) {
        if (src + 1 > avail) return false;
        const bool final_block = z[src] & 1;
        if (((z[src] >> 1) & 3) != 0) return false;
        src++;
        if (src + 4 > avail) return false;
        const uint32_t len = z[src] | (z[src + 1] << 8), nlen = z[src + 2] | (z[src + 3] << 8);
        src += 4;
        if (len != (~nlen & 0xFFFF)) return false;
        if (src + len > avail) return false;
        for (uint32_t i = 0; i < len; i++) {
            const uint8_t c = z[src + i];
            if (!raster) {
                if (c != 0) return false; // stored files use filter 0 on every row
            } else {
                if (comp < dst_chans) {
                    if (out == dst_len) return false;
                    dst[out++] = c;
                }
                if (++comp == src_chans) {
                    if (dst_chans > src_chans) {
                        if (out == dst_len) return false;
                        dst[out++] = 0xFF;
                    }
                    comp = 0;
                }
            }
            if (++raster == src_bpl + 1) raster = 0;
        }
        src += len;
        if (final_block) break;
    }
    if (comp) return false;
    if (src + 4 != zlib_len) return false;
    return out == dst_len;
}

bool inflate_pixels(const uint8_t *z, uint32_t avail, uint32_t zlib_len, uint8_t *dst, uint32_t w, uint32_t h, uint32_t src_chans,
                    uint32_t dst_chans)
{
    if (zlib_len < 7) return false;
    if (z[0] != 0x78 || z[1] != 0x01) return false;
    if ((z[2] & 6) == 0) return inflate_stored(z, avail, zlib_len, dst, w, h, src_chans, dst_chans);
    Bits in = {z, avail, 2, 0, 0, false};
    if (in.get(1) != 1 || in.get(2) != 2) return false; // one final dynamic block
    static thread_local uint32_t lit_table[1u << kTableBits];
    if (!read_dynamic_header(in, src_chans, lit_table)) return false;

    static const uint16_t len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                          31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const size_t dst_bpl = (size_t)w * dst_chans;
    const uint8_t *prev_row = nullptr;
    uint8_t *row = dst;
    for (uint32_t y = 0; y < h; y++) {
        uint32_t e = lit_table[in.peek(kTableBits)];
        if (!(e >> 9)) return false;
        in.skip(e >> 9);
        if ((e & 511) != (y ? 2u : 0u)) return false; // filter type literal
        uint8_t delta[4] = {0, 0, 0, 0};             // previous pixel in FILTERED space
        uint32_t x = 0;
        while (x < w) {
            e = lit_table[in.peek(kTableBits)];
            if (!(e >> 9)) return false;
            in.skip(e >> 9);
            uint32_t sym = e & 511, npix = 1;
            if (sym & 256) {
                if (sym == 256 || sym > 285) return false; // EOB with pixels left, or not a length symbol
                uint32_t run = len_base[sym - 257];
                if (len_extra[sym - 257]) run += in.get(len_extra[sym - 257]);
                in.skip(1); // distance code: always the 1-bit code of "previous pixel"
                if (run % src_chans) return false;
                npix = run / src_chans;
                if (!npix || x + npix > w) return false; // whole pixels, inside the row
            } else {
                delta[0] = (uint8_t)sym;
                for (uint32_t k = 1; k < src_chans; k++) {
                    e = lit_table[in.peek(kTableBits)];
                    if (!(e >> 9)) return false;
                    in.skip(e >> 9);
                    if (e & 256) return false; // a pixel is never split by a match
                    delta[k] = (uint8_t)(e & 255);
                }
            }
            for (uint32_t i = 0; i < npix; i++, x++) {
                uint8_t *o = row + (size_t)x * dst_chans;
                const uint8_t *u = prev_row ? prev_row + (size_t)x * dst_chans : nullptr;
                o[0] = (uint8_t)((u ? u[0] : 0) + delta[0]);
                o[1] = (uint8_t)((u ? u[1] : 0) + delta[1]);
                o[2] = (uint8_t)((u ? u[2] : 0) + delta[2]);
                if (dst_chans == 4) o[3] = (src_chans == 4) ? (uint8_t)((u ? u[3] : 0) + delta[3]) : 0xFF;
            }
        }
        prev_row = row;
        row += dst_bpl;
    }
    const uint32_t e = lit_table[in.peek(kTableBits)];
    if (!(e >> 9) || (e & 511) != 256) return false;
    in.skip(e >> 9);
    const size_t end_byte = (in.bitpos() + 7) >> 3;
    return end_byte + 4 == zlib_len;
}

} // namespace

int fpng_get_info(const void *pImage, uint32_t image_size, uint32_t &width, uint32_t &height, uint32_t &channels_in_file)
{
    uint32_t o = 0, l = 0;
    if (!pImage) {
        width = height = channels_in_file = 0;
        return FPNG_DECODE_FAILED_NOT_PNG;
    }
    return parse_container(static_cast<const uint8_t *>(pImage), image_size, width, height, channels_in_file, o, l);
}

int fpng_decode_memory(const void *pImage, uint32_t image_size, std::vector<uint8_t> &out, uint32_t &width, uint32_t &height,
                       uint32_t &channels_in_file, uint32_t desired_channels)
{
    // (the reference empties `out` first and sizes it later, reference src/fpng.cpp:3087-3105: a vector reused from call to call
    // is zero-filled every time.  Here it is emptied on the ways out that fail and otherwise resized once, so that a reused
    // vector of the right size is only written by the pixels.)
    width = height = channels_in_file = 0;
    if (!pImage || !image_size || (desired_channels != 3 && desired_channels != 4)) {
        out.resize(0);
        return FPNG_DECODE_INVALID_ARG;
    }
    const uint8_t *png = static_cast<const uint8_t *>(pImage);
    uint32_t idat_ofs = 0, idat_len = 0;
    const int st = parse_container(png, image_size, width, height, channels_in_file, idat_ofs, idat_len);
    if (st) {
        out.resize(0);
        return st;
    }
    const uint64_t need = (uint64_t)width * height * desired_channels;
    if (need > UINT32_MAX) {
        out.resize(0);
        return FPNG_DECODE_FAILED_DIMENSIONS_TOO_LARGE;
    }
    // GPU tier: large images (the fixed costs of a GPU decode -- a dozen launches, two copies -- are ~0.4 ms; the CPU decoder below
    // does 80-400 MP/s).  FPNG_AMD_DECODE_CPU=1 keeps everything on the CPU.  Same pixels, same status codes (tests/test_gpu_decode.py);
    // a file the GPU path leaves undecided (token boundaries that do not synchronise) falls through to the CPU decoder.
    static const bool cpu_only = [] {
        const char *v = getenv("FPNG_AMD_DECODE_CPU");
        return v && v[0] == '1';
    }();
    if (!cpu_only && (uint64_t)width * height >= kGpuDecodeMinPixels) {
        if (fpng_amd_encoder *enc = dropin_thread_encoder()) {
            fpng_amd_decode_result r;
            const int rc = fpng_amd_decode_host(enc, pImage, image_size, desired_channels,
                                                [](void *user, size_t bytes) -> uint8_t * {
                                                    auto *v = static_cast<std::vector<uint8_t> *>(user);
                                                    v->resize(bytes);
                                                    return v->data();
                                                },
                                                &out, &r);
            if (rc == FPNG_AMD_OK && r.status != FPNG_AMD_DECODE_UNDECIDED) {
                if (r.status) out.resize(0);
                g_gpu_decodes.fetch_add(1, std::memory_order_relaxed);
                return r.status;
            }
            out.resize(0);
        }
    }
    out.resize((size_t)need);
    const uint8_t *z = png + idat_ofs + 8;
    const uint32_t avail = image_size - (idat_ofs + 8);
    // a 4-channel file whose alpha deltas are dropped still needs them for the run logic: handled inside
    if (!inflate_pixels(z, avail, idat_len, out.data(), width, height, channels_in_file, desired_channels)) {
        out.resize(0); // (the reference leaves whatever was decoded so far; callers must not look at it)
        return FPNG_DECODE_NOT_FPNG;
    }
    return FPNG_DECODE_SUCCESS;
}

unsigned long long gpu_decodes() { return g_gpu_decodes.load(std::memory_order_relaxed); }

#ifndef FPNG_NO_STDIO
int fpng_decode_file(const char *pFilename, std::vector<uint8_t> &out, uint32_t &width, uint32_t &height, uint32_t &channels_in_file,
                     uint32_t desired_channels)
{
    FILE *f = fopen(pFilename, "rb");
    if (!f) return FPNG_DECODE_FILE_OPEN_FAILED;
    if (fseek(f, 0, SEEK_END) != 0) {
        fclose(f);
        return FPNG_DECODE_FILE_SEEK_FAILED;
    }
    const long long size = ftello(f);
    if (fseek(f, 0, SEEK_SET) != 0) {
        fclose(f);
        return FPNG_DECODE_FILE_SEEK_FAILED;
    }
    if (size < 0 || size > (long long)UINT32_MAX) {
        fclose(f);
        return FPNG_DECODE_FILE_TOO_LARGE;
    }
    std::vector<uint8_t> buf((size_t)size);
    if (fread(buf.data(), 1, buf.size(), f) != buf.size()) {
        fclose(f);
        return FPNG_DECODE_FILE_READ_FAILED;
    }
    fclose(f);
    return fpng_decode_memory(buf.data(), (uint32_t)buf.size(), out, width, height, channels_in_file, desired_channels);
}
#endif

} // namespace fpng

// (not part of the reference's interface) calls of fpng_decode_memory / fpng_decode_file in this process that the GPU tier answered
extern "C" unsigned long long fpng_amd_dropin_gpu_decodes() { return fpng::gpu_decodes(); }
