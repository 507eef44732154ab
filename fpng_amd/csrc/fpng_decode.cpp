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
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace fpng {

fpng_amd_encoder *dropin_thread_encoder(); // fpng_dropin.cpp

namespace {

constexpr uint64_t kGpuDecodeMinPixels = 1u << 18; // images of 512 x 512 pixels and more are decoded on the GPU
std::atomic<uint64_t> g_gpu_decodes{0};

using namespace parse;

// Stored blocks (reference src/fpng.cpp:2107-2207): the filtered bytes -- filter 0 on every row -- lie in the blocks' payloads.  The
// reference walks them a byte at a time; here whole rows are copied.  What it accepts, kept: every block header up to the final
// block must be sound (type 0, LEN = ~NLEN, payload inside the data) whether or not its bytes are wanted; the rows' filter bytes are
// 0; the payloads hold the image exactly -- or ONE more byte if it is 0 (read as the filter byte of a row that never comes: the
// reference looks at a row's first byte before it asks whether there is room); the Adler-32 follows the final block.
struct StoredBlocks {
    const uint8_t *z;
    uint64_t src, avail;
    uint32_t left = 0; // payload bytes of the current block not taken yet
    bool final_seen = false, bad = false;
    // -> bytes copied (fewer than n: the final block is used up, or a header is bad)
    size_t read(uint8_t *dst, size_t n)
    {
        size_t got = 0;
        while (got < n) {
            if (!left) {
                if (final_seen || bad) break;
                if (src + 1 > avail || ((z[src] >> 1) & 3) != 0 || src + 5 > avail) {
                    bad = true;
                    break;
                }
                final_seen = z[src] & 1;
                const uint32_t len = z[src + 1] | (z[src + 2] << 8), nlen = z[src + 3] | (z[src + 4] << 8);
                src += 5;
                if (len != (~nlen & 0xFFFFu) || src + len > avail) {
                    bad = true;
                    break;
                }
                left = len;
                continue;
            }
            const size_t k = n - got < left ? n - got : left;
            memcpy(dst + got, z + src, k);
            got += k, src += k, left -= (uint32_t)k;
        }
        return got;
    }
};

bool inflate_stored(const uint8_t *z, uint32_t avail, uint32_t zlib_len, uint8_t *dst, uint32_t w, uint32_t h, uint32_t src_chans,
                    uint32_t dst_chans)
{
    const size_t src_bpl = (size_t)w * src_chans, dst_bpl = (size_t)w * dst_chans;
    StoredBlocks in = {z, 2, avail};
    std::vector<uint8_t> tmp(src_chans == dst_chans ? 0 : src_bpl + 4);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t f;
        if (in.read(&f, 1) != 1 || f != 0) return false; // stored files use filter 0 on every row
        uint8_t *o = dst + (size_t)y * dst_bpl;
        if (src_chans == dst_chans) {
            if (in.read(o, src_bpl) != src_bpl) return false;
        } else {
            if (in.read(tmp.data(), src_bpl) != src_bpl) return false;
            const uint8_t *t = tmp.data();
            if (dst_chans == 4)
                for (uint32_t x = 0; x < w; x++, t += 3, o += 4) { // (one 32-bit load -- the row buffer has the slack --, one store)
                    uint32_t v;
                    memcpy(&v, t, 4);
                    v |= 0xFF000000u;
                    memcpy(o, &v, 4);
                }
            else
                for (uint32_t x = 0; x < w; x++, t += 4, o += 3) o[0] = t[0], o[1] = t[1], o[2] = t[2];
        }
    }
    uint8_t extra[2];
    const size_t more = in.read(extra, 2); // (walks through whatever blocks follow, empty ones included, up to the final one)
    if (in.bad || !in.final_seen || in.left) return false;
    if (more > 1 || (more == 1 && extra[0] != 0)) return false;
    return in.src + 4 == zlib_len;
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

// ---- one final dynamic block: rows of (filter literal, pixels) ----
// The rows are decoded as what they are to Deflate -- a stream of BYTES -- into a row buffer, then un-filtered with packed byte adds
// (unfilter_row): a lookup yields up to three literals at once (as many whole literal codes as fit the 12 index bits; the
// GPU decoder's table idea, decode_core.h), stored with one unaligned 32-bit write; a match repeats the previous FILTERED pixel.  The
// reference's rules, stated on byte positions (col = place in the row, 0 = the filter literal): a length symbol (or the end of the
// block) may only stand where a pixel starts -- "a pixel is never split by a match" --, a match is whole pixels and ends inside its
// row, the filter literal is 0 then 2, nothing follows the last row but the end-of-block symbol.
//   entry: literals: bits 23..0 the bytes (first one lowest), 25..24 their number n (1..3), 29..26 code bits of the whole group
//          n == 0: bit 12 match (8..0 base length, 11..9 extra bits), bit 13 end of block, bit 14 reserved length symbol (286 / 287);
//          29..26 the symbol's code bits; 0 = no such code
enum : uint32_t { kRowMatch = 1u << 12, kRowEob = 1u << 13, kRowReserved = 1u << 14 };
void build_row_table(const uint32_t *tab, uint32_t *rt)
{
    static const uint16_t len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                          31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (uint32_t k = 0; k < (1u << kTableBits); k++) {
        const uint32_t e1 = tab[k], l1 = e1 >> 9, s1 = e1 & 511u;
        uint32_t ent = 0;
        if (!l1)
            ent = 0;
        else if (s1 == 256)
            ent = l1 << 26 | kRowEob;
        else if (s1 > 285)
            ent = l1 << 26 | kRowReserved;
        else if (s1 > 256)
            ent = l1 << 26 | kRowMatch | (uint32_t)len_extra[s1 - 257] << 9 | len_base[s1 - 257];
        else {
            uint32_t L = l1, n = 1, lits = s1;
            while (n < 3) {
                const uint32_t e2 = tab[k >> L], l2 = e2 >> 9, s2 = e2 & 511u;
                if (!l2 || s2 >= 256 || L + l2 > kTableBits) break;
                lits |= s2 << (8 * n);
                n++, L += l2;
            }
            ent = L << 26 | n << 24 | lits;
        }
        rt[k] = ent;
    }
}

// out = up + f, byte-wise (the Up filter undone; row 0 has filter 0 and a row of zeros above it), SC channels in, DC channels out.
// Packed byte adds in 64-bit words (the compilers at hand do not vectorise the plain byte loop: 2.8 cycles a byte, more than the
// Huffman decoding in front of it).
inline uint64_t add_bytes64(uint64_t a, uint64_t b) { return ((a & 0x7F7F7F7F7F7F7F7Full) + (b & 0x7F7F7F7F7F7F7F7Full)) ^ ((a ^ b) & 0x8080808080808080ull); }
inline uint32_t add_bytes32(uint32_t a, uint32_t b) { return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u); }
template <int SC, int DC> void unfilter_row(const uint8_t *f, const uint8_t *up, uint8_t *out, uint32_t w)
{
    if (SC == DC) {
        const size_t n = (size_t)w * SC;
        size_t i = 0;
#if defined(__SSE2__)
        for (; i + 16 <= n; i += 16)
            _mm_storeu_si128((__m128i *)(out + i), _mm_add_epi8(_mm_loadu_si128((const __m128i *)(up + i)), _mm_loadu_si128((const __m128i *)(f + i))));
#endif
        for (; i + 8 <= n; i += 8) {
            uint64_t a, b;
            memcpy(&a, up + i, 8), memcpy(&b, f + i, 8);
            a = add_bytes64(a, b);
            memcpy(out + i, &a, 8);
        }
        for (; i < n; i++) out[i] = (uint8_t)(up[i] + f[i]);
    } else if (SC == 3) { // -> RGBA, alpha 0xFF: one 32-bit load (the fourth byte is the next pixel's, or slack), one store a pixel
        for (uint32_t x = 0; x < w; x++) {
            uint32_t a, b;
            memcpy(&a, up + 4 * (size_t)x, 4), memcpy(&b, f + 3 * (size_t)x, 4);
            a = add_bytes32(a, b) | 0xFF000000u;
            memcpy(out + 4 * (size_t)x, &a, 4);
        }
    } else { // RGBA -> RGB: the alpha deltas only ever mattered to the decoder's rules
        for (uint32_t x = 0; x < w; x++) {
            uint32_t b;
            memcpy(&b, f + 4 * (size_t)x, 4);
            out[3 * (size_t)x] = (uint8_t)(up[3 * (size_t)x] + b);
            out[3 * (size_t)x + 1] = (uint8_t)(up[3 * (size_t)x + 1] + (b >> 8));
            out[3 * (size_t)x + 2] = (uint8_t)(up[3 * (size_t)x + 2] + (b >> 16));
        }
    }
}

// SC channels in the file, DC channels out
template <int SC, int DC> bool inflate_rows(const FastBits &in_, const uint32_t *rt, uint8_t *dst, uint32_t w, uint32_t h, size_t *end_bit)
{
    FastBits in = in_; // (a local copy: its fields live in registers, byte stores into the row buffer cannot alias them)
    const size_t bpl = (size_t)w * SC, stride = bpl + 1, dst_bpl = (size_t)w * DC;
    std::vector<uint8_t> rows(2 * (stride + 16), 0), zero_row(dst_bpl, 0);
    uint8_t *rb = rows.data(), *nb = rows.data() + stride + 16; // this row's filtered bytes (filter literal first), the next row's
    const uint8_t *up = zero_row.data();
    uint8_t *out = dst;
    size_t fill = 0; // bytes of the current row that are there (a group of literals may have run over the previous row's end)
    size_t zero_run_bytes = 0; // ... of which runs of zero deltas (a row that is nothing else repeats the row above: flat content)
    for (uint32_t y = 0; y < h; y++) {
        while (fill < stride) {
            in.refill(); // >= 56 bits: three tokens (a match: 12 + 5 + 1 bits at most)
            if (in.byte > in.n + 16) return false; // far behind the end of the data (zeros from there on): the reference's reader has given up long ago
            for (int k = 0; k < 3 && fill < stride; k++) {
                const uint32_t e = rt[in.buf & 4095u], L = e >> 26, n = (e >> 24) & 3u;
                if (n) {
                    const uint32_t v = e & 0xFFFFFFu;
                    memcpy(rb + fill, &v, 4); // (one byte of slack behind n <= 3)
                    fill += n;
                    in.consume(L);
                    continue;
                }
                if (!L || (e & kRowEob)) return false; // no such code, or the block ends with pixels left
                // a length symbol: only where a pixel starts
                if (fill < 1 || (fill - 1) % SC) return false;
                in.consume(L);
                uint8_t *o = rb + fill;
                if (e & kRowReserved) {
                    // 286 / 287, the length symbols Deflate reserves (a hand-made table can give them codes): the reference's 3-channel
                    // decoder turns them away; its 4-channel decoder takes them for a match of length ZERO, and its copy loops run once
                    // before they ask (src/fpng.cpp:2668-2760): nothing happens where the previous pixel's deltas are all zero and there
                    // is a row above, elsewhere ONE more pixel is written.  No fpng encoder writes such a file; the answer is the reference's.
                    if (SC != 4) return false;
                    in.consume(1); // the distance code
                    uint32_t prev = 0;
                    if (fill > 1) memcpy(&prev, o - 4, 4);
                    if (y != 0 && !prev) continue;
                    memcpy(o, &prev, 4);
                    fill += 4;
                    continue;
                }
                const uint32_t xb = (e >> 9) & 7u;
                const uint32_t run = (e & 511u) + (uint32_t)(in.buf & ((1u << xb) - 1u));
                in.consume(xb + 1); // extra bits + the distance code: always the 1-bit code of "previous pixel"
                if (run % SC || fill + run > stride) return false; // whole pixels, inside the row
                // the previous FILTERED pixel, `run` bytes of it (the pixel in front of a row's first one is all zeros: reference :2268);
                // flat content is runs of zero deltas
                uint32_t px = 0;
                if (fill > 1) memcpy(&px, o - 4, 4), px = SC == 4 ? px : px >> 8;
                // (zero deltas, counted for the flat-row short cut below; a run at the row's first pixel -- no fpng encoder writes
                //  one -- is left out: the test below looks at that pixel itself)
                zero_run_bytes += (px || fill == 1) ? 0u : run;
                bool filled = false;
#if defined(__SSE2__)
                if (!px || SC == 4) { // 16 bytes a store, written in line (a run has 258 bytes at most; the row buffer has 16 bytes of slack)
                    const __m128i v = _mm_set1_epi32((int)px);
                    for (uint32_t i = 0; i < run; i += 16) _mm_storeu_si128((__m128i *)(o + i), v);
                    filled = true;
                }
#else
                if (!px) {
                    if (run <= 16)
                        memset(o, 0, 16); // (two inline stores; the row buffer has 16 bytes of slack)
                    else
                        memset(o, 0, run);
                    filled = true;
                } else if (SC == 4) {
                    const uint64_t p2 = (uint64_t)px << 32 | px;
                    for (uint32_t i = 0; i < run; i += 8) memcpy(o + i, &p2, 8); // (run is a multiple of 4; up to 4 bytes of slack behind the row)
                    filled = true;
                }
#endif
                if (!filled) {
                    // 3-byte pixels: 12 bytes = 4 pixels at a time from three rotated words
                    const uint64_t wrap = (uint64_t)px | (uint64_t)px << 24 | (uint64_t)px << 48;
                    const uint32_t w0 = (uint32_t)wrap, w1 = (uint32_t)(wrap >> 8), w2 = (uint32_t)(wrap >> 16);
                    for (uint32_t i = 0; i < run; i += 12) memcpy(o + i, &w0, 4), memcpy(o + i + 4, &w1, 4), memcpy(o + i + 8, &w2, 4); // (up to 9 bytes of slack)
                }
                fill += run;
            }
        }
        if (rb[0] != (y ? 2u : 0u)) return false; // filter type literal
        const size_t over = fill - stride;        // literals of the next row (at most two)
        if (over && y + 1 == h) return false;
        nb[0] = rb[stride], nb[1] = rb[stride + 1];
        // ---- Up filter undone (row 0: filter 0 = the bytes themselves; `up` is a row of zeros there), channels converted ----
        // (a flat row as fpng's encoders write it: the first pixel as literals -- no match stands there --, the rest runs of zero deltas)
        uint32_t first_px;
        memcpy(&first_px, rb + 1, 4);
        if (y && zero_run_bytes + SC == bpl && !(SC == 4 ? first_px : first_px & 0xFFFFFFu))
            memcpy(out, up, dst_bpl); // every delta of the row is zero (the reference's own short cut: src/fpng.cpp:2312-2316 copies the row above)
        else
            unfilter_row<SC, DC>(rb + 1, up, out, w);
        zero_run_bytes = 0;
        up = out;
        out += dst_bpl;
        std::swap(rb, nb);
        fill = over;
    }
    in.refill();
    const uint32_t e = rt[in.buf & 4095u];
    if (!(e >> 26) || ((e >> 24) & 3u) || !(e & kRowEob)) return false;
    in.consume(e >> 26);
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
    // (one heap block for both tables: thread-local arrays cost a __tls_get_addr call at every use in a shared library, and the
    //  compiler keeps some of them inside the row loop -- 45 % slower)
    std::vector<uint32_t> tables(2u << kTableBits);
    uint32_t *const lit_table = tables.data(), *const row_table = tables.data() + (1u << kTableBits);
    if (!read_dynamic_header(in, src_chans, lit_table)) return false;
    build_row_table(lit_table, row_table);
    FastBits fb = {in.p, in.n, in.byte, in.buf, in.cnt};
    size_t end_bit = 0;
    bool ok;
    if (src_chans == 3)
        ok = dst_chans == 3 ? inflate_rows<3, 3>(fb, row_table, dst, w, h, &end_bit) : inflate_rows<3, 4>(fb, row_table, dst, w, h, &end_bit);
    else
        ok = dst_chans == 3 ? inflate_rows<4, 3>(fb, row_table, dst, w, h, &end_bit) : inflate_rows<4, 4>(fb, row_table, dst, w, h, &end_bit);
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
    // The reference empties `out` first and sizes it to width * height * desired_channels once the container is accepted
    // (src/fpng.cpp:3087-3111): a call that fails before that point leaves an empty vector, one that fails in the stream
    // (FPNG_DECODE_NOT_FPNG from the inflater, :3131-3136) leaves a vector of the image's size whose contents are unspecified.
    // The same sizes leave here on every exit (tests/test_dropin_decode.py::test_vector_size_on_every_exit); what is not
    // repeated is the zero fill of a reused vector on every call (resize(0) + resize(n) writes n zeros: 133 MB for an 8K frame).
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
                out.resize((size_t)need); // (a stream that fails leaves the sized vector, like the reference)
                g_gpu_decodes.fetch_add(1, std::memory_order_relaxed);
                return r.status;
            }
        }
    }
    out.resize((size_t)need);
    const uint8_t *z = png + idat_ofs + 8;
    const uint32_t avail = image_size - (idat_ofs + 8);
    // a 4-channel file whose alpha deltas are dropped still needs them for the run logic: handled inside
    if (!inflate_pixels(z, avail, idat_len, out.data(), width, height, channels_in_file, desired_channels)) {
        return FPNG_DECODE_NOT_FPNG; // (`out` keeps the image's size, its contents are whatever was decoded so far: src/fpng.cpp:3131-3136)
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
