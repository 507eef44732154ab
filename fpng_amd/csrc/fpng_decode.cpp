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
#include "fpng.h"

#include "fpng_amd.h"

#include <stdio.h>
#include <string.h>

namespace fpng {

namespace {

const uint32_t kMaxDim = 1u << 24;

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int parse_container(const uint8_t *png, uint32_t size, uint32_t &w, uint32_t &h, uint32_t &chans, uint32_t &idat_ofs,
                    uint32_t &idat_len)
{
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    w = h = chans = idat_ofs = idat_len = 0;
    // signature + IHDR chunk (25) + chunk prefix (8) + 1 + crc (4) + IEND (12)
    if (size < 8 + 25 + 8 + 1 + 4 + 12) return FPNG_DECODE_FAILED_NOT_PNG;
    if (memcmp(png, sig, 8) != 0) return FPNG_DECODE_FAILED_NOT_PNG;
    const uint8_t *ihdr = png + 8;
    if (be32(ihdr) != 13) return FPNG_DECODE_FAILED_NOT_PNG;
    if (fpng_amd_crc32(ihdr + 4, 4 + 13, 0) != be32(ihdr + 21)) return FPNG_DECODE_FAILED_HEADER_CRC32;
    w = be32(ihdr + 8);
    h = be32(ihdr + 12);
    if (!w || !h || w > kMaxDim || h > kMaxDim) return FPNG_DECODE_FAILED_INVALID_DIMENSIONS;
    if ((uint64_t)w * h > (1u << 30)) return FPNG_DECODE_FAILED_INVALID_DIMENSIONS;
    if (ihdr[18] || ihdr[19] || ihdr[20] || ihdr[16] != 8) return FPNG_DECODE_NOT_FPNG;
    if (ihdr[17] == 2)
        chans = 3;
    else if (ihdr[17] == 6)
        chans = 4;
    else
        return FPNG_DECODE_NOT_FPNG;

    bool have_fdec = false;
    size_t ofs = 8 + 25;
    for (;;) {
        if (ofs >= size) return FPNG_DECODE_FAILED_CHUNK_PARSING;
        if (size - ofs < 12) return FPNG_DECODE_FAILED_CHUNK_PARSING;
        const uint8_t *ck = png + ofs;
        const uint32_t len = be32(ck);
        if (ofs + 8 + (uint64_t)len + 4 > size) return FPNG_DECODE_FAILED_CHUNK_PARSING;
        for (int i = 0; i < 4; i++) {
            const uint8_t c = ck[4 + i];
            if (!((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) return FPNG_DECODE_FAILED_CHUNK_PARSING;
        }
        const bool is_idat = memcmp(ck + 4, "IDAT", 4) == 0;
        if (!is_idat && fpng_amd_crc32(ck + 4, 4 + len, 0) != be32(ck + 8 + len)) return FPNG_DECODE_FAILED_HEADER_CRC32;
        const uint8_t *data = ck + 8;
        if (memcmp(ck + 4, "IEND", 4) == 0) break;
        if (is_idat) {
            if (idat_ofs || !have_fdec) return FPNG_DECODE_NOT_FPNG; // second IDAT, or IDAT before the marker
            idat_ofs = (uint32_t)ofs;
            idat_len = len;
            if (idat_len < 7) return FPNG_DECODE_FAILED_INVALID_IDAT;
        } else if (memcmp(ck + 4, "fdEC", 4) == 0) {
            if (have_fdec || len != 5) return FPNG_DECODE_NOT_FPNG;
            if (data[0] != 82 || data[1] != 36 || data[2] != 147 || data[3] != 227 || data[4] != 0) return FPNG_DECODE_NOT_FPNG;
            have_fdec = true;
        } else if ((ck[4] & 32) == 0) {
            return FPNG_DECODE_NOT_FPNG; // unknown critical chunk
        }
        ofs += 8 + (size_t)len + 4;
    }
    if (!have_fdec || !idat_ofs) return FPNG_DECODE_NOT_FPNG;
    return FPNG_DECODE_SUCCESS;
}

// LSB-first bit reader confined to the zlib payload
struct Bits {
    const uint8_t *p;
    size_t n;      // bytes available
    size_t byte;   // next byte to load
    uint64_t buf;
    uint32_t cnt;
    bool overrun;
    void fill()
    {
        while (cnt <= 56) {
            uint64_t b = 0;
            if (byte < n)
                b = p[byte];
            byte++;
            buf |= b << cnt;
            cnt += 8;
        }
    }
    uint32_t peek(uint32_t k)
    {
        if (cnt < k) fill();
        return (uint32_t)(buf & ((1ull << k) - 1));
    }
    void skip(uint32_t k)
    {
        if (cnt < k) fill();
        buf >>= k;
        cnt -= k;
    }
    uint32_t get(uint32_t k)
    {
        const uint32_t v = peek(k);
        skip(k);
        return v;
    }
    // position of the next unread bit
    size_t bitpos() const { return byte * 8 - cnt; }
};

const uint32_t kTableBits = 12;

// 12-bit direct lookup: sym | len << 9 (0 = invalid)
bool build_lookup(const uint8_t *len, uint32_t n, uint32_t *table)
{
    uint32_t per_len[16] = {0};
    for (uint32_t i = 0; i < n; i++) per_len[len[i]]++;
    per_len[0] = 0;
    uint32_t kraft = 0, used = 0;
    for (uint32_t l = 1; l <= 15; l++) {
        kraft += per_len[l] << (15 - l);
        used += per_len[l];
    }
    if (kraft != (1u << 15) && used != 1) return false; // complete code, or the single-code special case
    uint32_t first[16] = {0}, code = 0;
    for (uint32_t l = 1; l <= 15; l++) {
        code = (code + per_len[l - 1]) << 1;
        first[l] = code;
    }
    memset(table, 0, sizeof(uint32_t) << kTableBits);
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t l = len[s];
        if (!l) continue;
        uint32_t c = first[l]++, r = 0;
        for (uint32_t i = 0; i < l; i++, c >>= 1) r = (r << 1) | (c & 1);
        for (; r < (1u << kTableBits); r += 1u << l) table[r] = s | (l << 9);
    }
    return true;
}

bool read_dynamic_header(Bits &in, uint32_t chans, uint32_t *lit_table)
{
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const uint32_t n_lit = in.get(5) + 257, n_dist = in.get(5) + 1, total = n_lit + n_dist;
    if (total > 288 + 32) return false;
    const uint32_t n_clc = in.get(4) + 4;
    uint8_t clc[19] = {0};
    for (uint32_t i = 0; i < n_clc; i++) clc[order[i]] = (uint8_t)in.get(3);
    static thread_local uint32_t clc_table[1u << kTableBits];
    if (!build_lookup(clc, 19, clc_table)) return false;
    uint8_t sizes[288 + 32];
    memset(sizes, 0, sizeof sizes);
    for (uint32_t cur = 0; cur < total;) {
        const uint32_t e = clc_table[in.peek(kTableBits)];
        if (!(e >> 9)) return false;
        in.skip(e >> 9);
        const uint32_t sym = e & 511;
        if (sym <= 15) {
            if (sym > kTableBits) return false; // fpng never emits codes longer than 12 bits
            sizes[cur++] = (uint8_t)sym;
            continue;
        }
        uint32_t rep, val = 0;
        if (sym == 16) {
            rep = in.get(2) + 3;
            if (!cur) return false;
            val = sizes[cur - 1];
        } else if (sym == 17)
            rep = in.get(3) + 3;
        else
            rep = in.get(7) + 11;
        if (cur + rep > total) return false;
        while (rep--) sizes[cur++] = (uint8_t)val;
    }
    // distance tree: one or two 1-bit codes, the pixel distance among them
    uint32_t one_bit = 0;
    for (uint32_t i = 0; i < n_dist; i++) one_bit += sizes[n_lit + i] == 1;
    if (one_bit < 1 || one_bit > 2) return false;
    if (sizes[n_lit + chans - 1] != 1) return false;
    if (one_bit == 2 && sizes[n_lit + chans] != 1) return false;
    uint8_t lit[288];
    memcpy(lit, sizes, n_lit);
    memset(lit + n_lit, 0, 288 - n_lit);
    return build_lookup(lit, n_lit, lit_table);
}

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
    out.resize(0);
    width = height = channels_in_file = 0;
    if (!pImage || !image_size || (desired_channels != 3 && desired_channels != 4)) return FPNG_DECODE_INVALID_ARG;
    const uint8_t *png = static_cast<const uint8_t *>(pImage);
    uint32_t idat_ofs = 0, idat_len = 0;
    const int st = parse_container(png, image_size, width, height, channels_in_file, idat_ofs, idat_len);
    if (st) return st;
    const uint64_t need = (uint64_t)width * height * desired_channels;
    if (need > UINT32_MAX) return FPNG_DECODE_FAILED_DIMENSIONS_TOO_LARGE;
    out.resize((size_t)need);
    const uint8_t *z = png + idat_ofs + 8;
    const uint32_t avail = image_size - (idat_ofs + 8);
    // a 4-channel file whose alpha deltas are dropped still needs them for the run logic: handled inside
    if (!inflate_pixels(z, avail, idat_len, out.data(), width, height, channels_in_file, desired_channels))
        return FPNG_DECODE_NOT_FPNG;
    return FPNG_DECODE_SUCCESS;
}

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
