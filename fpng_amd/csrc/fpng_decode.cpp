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
#include <vector>

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

// The pixel loops' bit reader: like parse::Bits (bytes behind the input read as zero), refilled eight bytes at a time.  After
// refill() at least 56 bits are there: one pixel (four literals of <= 12 bits, or a length symbol + 5 extra bits + the distance bit)
// is decoded without another look at the fill level.
struct FastBits {
    const uint8_t *p;
    size_t n, byte;
    uint64_t buf;
    uint32_t cnt;
    inline void refill()
    {
        if (byte + 8 <= n) {
            uint64_t v;
            memcpy(&v, p + byte, 8); // (little-endian host, like everything else here)
            buf |= v << cnt;           // (bits above cnt are the stream's next bits already: OR-ing them in again next time is harmless)
            byte += (63u - cnt) >> 3; // whole bytes that fit
            cnt |= 56u;
        } else {
            while (cnt <= 56) {
                const uint64_t b8 = byte < n ? p[byte] : 0;
                byte++;
                buf |= b8 << cnt;
                cnt += 8;
            }
        }
    }
    inline void consume(uint32_t k)
    {
        buf >>= k;
        cnt -= k;
    }
    size_t bitpos() const { return byte * 8 - cnt; }
};

// one final dynamic block: rows of (filter literal, pixels); SC channels in the file, DC channels out
template <int SC, int DC> bool inflate_rows(FastBits &in, const uint32_t *tab, uint8_t *dst, uint32_t w, uint32_t h, size_t *end_bit)
{
    static const uint16_t len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                          31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const size_t dst_bpl = (size_t)w * DC;
    std::vector<uint8_t> zero_row(dst_bpl, 0); // "the row above" of row 0
    const uint8_t *up = zero_row.data();
    uint8_t *row = dst;
    for (uint32_t y = 0; y < h; y++) {
        in.refill();
        uint32_t e = tab[in.buf & 4095u];
        if (!(e >> 9)) return false;
        in.consume(e >> 9);
        if ((e & 511u) != (y ? 2u : 0u)) return false; // filter type literal
        uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;         // previous pixel in FILTERED space
        uint32_t x = 0;
        while (x < w) {
            in.refill();
            e = tab[in.buf & 4095u];
            if (!(e >> 9)) return false;
            in.consume(e >> 9);
            const uint32_t sym = e & 511u;
            uint32_t npix = 1;
            if (sym & 256u) {
                if (sym == 256u) return false; // EOB with pixels left
                if (sym > 285u) {
                    // 286 / 287, the length symbols Deflate reserves (a hand-made table can give them codes): the reference's 3-channel
                    // decoder turns them away; its 4-channel decoder takes them for a match of length ZERO, and its copy loops run once
                    // before they ask (src/fpng.cpp:2668-2760): nothing happens where the previous pixel's deltas are all zero and there
                    // is a row above, elsewhere ONE more pixel is written.  No fpng encoder writes such a file; the answer is the reference's.
                    if (SC != 4) return false;
                    in.consume(1); // the distance code
                    if (y != 0 && !(d0 | d1 | d2 | d3)) continue;
                    goto write_pixels; // (npix = 1)
                }
                const uint32_t xb = len_extra[sym - 257];
                const uint32_t run = len_base[sym - 257] + (uint32_t)(in.buf & ((1u << xb) - 1u));
                in.consume(xb + 1); // extra bits + the distance code: always the 1-bit code of "previous pixel"
                if (run % SC) return false;
                npix = run / SC;
                if (!npix || x + npix > w) return false; // whole pixels, inside the row
            } else {
                d0 = sym;
                e = tab[in.buf & 4095u];
                if (!(e >> 9) || (e & 256u)) return false; // (a pixel is never split by a match)
                in.consume(e >> 9);
                d1 = e & 255u;
                e = tab[in.buf & 4095u];
                if (!(e >> 9) || (e & 256u)) return false;
                in.consume(e >> 9);
                d2 = e & 255u;
                if (SC == 4) {
                    e = tab[in.buf & 4095u];
                    if (!(e >> 9) || (e & 256u)) return false;
                    in.consume(e >> 9);
                    d3 = e & 255u;
                }
            }
        write_pixels:
            uint8_t *o = row + (size_t)x * DC;
            const uint8_t *u = up + (size_t)x * DC;
            if (DC == 4) {
                // four byte-wise sums in one 32-bit operation (SWAR); an RGB file's alpha is 0xFF whatever the row above holds
                const uint32_t d = d0 | (d1 << 8) | (d2 << 16) | (d3 << 24);
                for (uint32_t i = 0; i < npix; i++, o += 4, u += 4) {
                    uint32_t a;
                    memcpy(&a, u, 4);
                    uint32_t s = ((a & 0x7F7F7F7Fu) + (d & 0x7F7F7F7Fu)) ^ ((a ^ d) & 0x80808080u);
                    if (SC == 3) s |= 0xFF000000u;
                    memcpy(o, &s, 4);
                }
            } else {
                for (uint32_t i = 0; i < npix; i++, o += 3, u += 3) {
                    o[0] = (uint8_t)(u[0] + d0);
                    o[1] = (uint8_t)(u[1] + d1);
                    o[2] = (uint8_t)(u[2] + d2);
                }
            }
            x += npix;
        }
        up = row;
        row += dst_bpl;
    }
    in.refill();
    const uint32_t e = tab[in.buf & 4095u];
    if (!(e >> 9) || (e & 511u) != 256u) return false;
    in.consume(e >> 9);
    *end_bit = in.bitpos();
    return true;
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
    FastBits fb = {in.p, in.n, in.byte, in.buf, in.cnt};
    size_t end_bit = 0;
    bool ok;
    if (src_chans == 3)
        ok = dst_chans == 3 ? inflate_rows<3, 3>(fb, lit_table, dst, w, h, &end_bit) : inflate_rows<3, 4>(fb, lit_table, dst, w, h, &end_bit);
    else
        ok = dst_chans == 3 ? inflate_rows<4, 3>(fb, lit_table, dst, w, h, &end_bit) : inflate_rows<4, 4>(fb, lit_table, dst, w, h, &end_bit);
    if (!ok) return false;
    return ((end_bit + 7) >> 3) + 4 == zlib_len;
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
