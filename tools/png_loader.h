// png_loader.h -- a small, self-contained PNG reader for the fpng_amd_test harness: ANY PNG, interlaced (Adam7) or not (grey, RGB,
// palette, grey+alpha, RGBA; 1/2/4/8/16 bits per sample; tRNS) to 8-bit RGBA, the role lodepng_decode_memory(..., LCT_RGBA, 8)
// plays in the reference's harness (reference src/fpng_test.cpp:1116-1122).  Plain RFC 1950/1951 inflate + PNG filters 0-4
// (RFC 2083), no dependencies.  Test-tool code: the product never reads PNGs other than its own (fpng_decode.cpp).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

namespace png_loader {

struct BitReader {
    const uint8_t *p;
    size_t n, pos = 0;
    uint64_t acc = 0;
    int cnt = 0;
    bool ok = true;
    BitReader(const uint8_t *d, size_t len) : p(d), n(len) {}
    void fill()
    {
        while (cnt <= 56 && pos < n) acc |= (uint64_t)p[pos++] << cnt, cnt += 8;
    }
    uint32_t bits(int k)
    {
        if (!k) return 0;
        if (cnt < k) fill();
        if (cnt < k) {
            ok = false;
            return 0;
        }
        const uint32_t v = (uint32_t)(acc & ((1ull << k) - 1));
        acc >>= k, cnt -= k;
        return v;
    }
    void align() { acc >>= (cnt & 7), cnt -= (cnt & 7); }
};

struct Huff { // canonical code, decoded bit by bit against per-length first-code tables
    uint16_t count[16], symbol[320];
    bool build(const uint8_t *len, int n)
    {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) {
            left = (left << 1) - count[l];
            if (left < 0) return false;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
        for (int i = 0; i < n; i++)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)br.bits(1);
            if (!br.ok) return -1;
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c, first += c, first <<= 1, code <<= 1;
        }
        return -1;
    }
};

inline bool inflate_zlib(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t expect)
{
    if (n < 6 || (src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 || (src[1] & 0x20)) return false;
    BitReader br(src + 2, n - 2);
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    out.clear();
    out.reserve(expect);
    for (bool last = false; !last;) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.align();
            const uint32_t len = br.bits(16), nlen = br.bits(16);
            if (!br.ok || (len ^ 0xFFFF) != nlen) return false;
            for (uint32_t i = 0; i < len; i++) {
                out.push_back((uint8_t)br.bits(8));
                if (!br.ok) return false;
            }
            continue;
        }
        if (type == 3) return false;
        Huff lit, dist;
        uint8_t lens[320];
        if (type == 1) {
            for (int i = 0; i < 288; i++) lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            lit.build(lens, 288);
            for (int i = 0; i < 30; i++) lens[i] = 5;
            dist.build(lens, 30);
        } else {
            const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
            Huff clh;
            if (!br.ok || nlen > 286 || ndist > 30 || !clh.build(cl, 19)) return false;
            for (int i = 0; i < nlen + ndist;) {
                const int sym = clh.decode(br);
                if (sym < 0) return false;
                if (sym < 16)
                    lens[i++] = (uint8_t)sym;
                else {
                    int rep;
                    uint8_t v = 0;
                    if (sym == 16) {
                        if (!i) return false;
                        v = lens[i - 1], rep = 3 + (int)br.bits(2);
                    } else
                        rep = sym == 17 ? 3 + (int)br.bits(3) : 11 + (int)br.bits(7);
                    if (i + rep > nlen + ndist) return false;
                    while (rep--) lens[i++] = v;
                }
            }
            if (!lit.build(lens, nlen) || !dist.build(lens + nlen, ndist)) return false;
        }
        for (;;) {
            const int sym = lit.decode(br);
            if (sym < 0) return false;
            if (sym < 256)
                out.push_back((uint8_t)sym);
            else if (sym == 256)
                break;
            else {
                if (sym > 285) return false;
                const uint32_t len = lbase[sym - 257] + br.bits(lext[sym - 257]);
                const int ds = dist.decode(br);
                if (ds < 0 || ds > 29) return false;
                const uint32_t d = dbase[ds] + br.bits(dext[ds]);
                if (!br.ok || d > out.size()) return false;
                const size_t from = out.size() - d;
                for (uint32_t i = 0; i < len; i++) out.push_back(out[from + i]);
            }
        }
    }
    return true; // (the Adler-32 is not checked: the chunk CRCs already were not either; this is a loader for test inputs)
}

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

// -> 8-bit RGBA, width, height; false with a reason in `err`
inline bool load_rgba(const uint8_t *d, size_t n, std::vector<uint8_t> &rgba, uint32_t &w, uint32_t &h, std::string &err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 33 || memcmp(d, sig, 8)) return err = "not a PNG file", false;
    uint32_t depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false;
    for (size_t o = 8; o + 12 <= n;) {
        const uint32_t len = be32(d + o);
        const uint8_t *type = d + o + 4, *body = d + o + 8;
        if ((size_t)len + 12 > n - o) return err = "truncated chunk", false;
        if (!memcmp(type, "IHDR", 4) && len == 13) {
            w = be32(body), h = be32(body + 4), depth = body[8], ctype = body[9], interlace = body[12];
            have_ihdr = true;
        } else if (!memcmp(type, "IDAT", 4))
            idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "PLTE", 4))
            plte.assign(body, body + len);
        else if (!memcmp(type, "tRNS", 4))
            trns.assign(body, body + len);
        else if (!memcmp(type, "IEND", 4))
            break;
        o += (size_t)len + 12;
    }
    if (!have_ihdr || !w || !h || (uint64_t)w * h > (1ull << 30)) return err = "bad IHDR", false;
    if (interlace > 1) return err = "unknown interlace method", false;
    const uint32_t chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) || ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!chans || !depth_ok) return err = "unsupported colour type / bit depth", false;
    const uint32_t bpp_bits = chans * depth, bpp = (bpp_bits + 7) / 8; // filter unit in bytes
    // the image as ONE pass, or as the seven passes of Adam7 (RFC 2083 section 2.6): sub-images of every dx-th column from x0 and
    // every dy-th row from y0, each filtered and stored like a small image of its own
    struct Pass {
        uint32_t x0, y0, dx, dy;
    };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass *passes = interlace ? adam7 : &whole;
    const int n_passes = interlace ? 7 : 1;
    size_t expect = 0;
    for (int q = 0; q < n_passes; q++) {
        const Pass &ps = passes[q];
        const uint32_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
        if (pw && ph) expect += (((size_t)pw * bpp_bits + 7) / 8 + 1) * ph;
    }
    std::vector<uint8_t> raw;
    if (!inflate_zlib(idat.data(), idat.size(), raw, expect) || raw.size() < expect) return err = "IDAT does not inflate to the image", false;
    rgba.resize((size_t)w * h * 4);
    const uint32_t maxv = (1u << (depth > 8 ? 8 : depth)) - 1;
    size_t at = 0;
    for (int q = 0; q < n_passes; q++) {
        const Pass &ps = passes[q];
        const uint32_t pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
        if (!pw || !ph) continue;
        const size_t stride = ((size_t)pw * bpp_bits + 7) / 8;
        uint8_t *base = raw.data() + at;
        at += (stride + 1) * ph;
        // ---- undo the row filters in place (RFC 2083 section 6) ----
        std::vector<uint8_t> zero(stride, 0);
        for (uint32_t y = 0; y < ph; y++) {
            uint8_t *row = base + (size_t)y * (stride + 1) + 1;
            const uint8_t *up = y ? row - (stride + 1) : zero.data();
            const uint32_t f = row[-1];
            if (f > 4) return err = "bad filter type", false;
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= bpp ? row[i - bpp] : 0, b = up[i], c = i >= bpp ? up[i - bpp] : 0;
                int pred = 0;
                if (f == 1)
                    pred = a;
                else if (f == 2)
                    pred = b;
                else if (f == 3)
                    pred = (a + b) >> 1;
                else if (f == 4) {
                    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                }
                row[i] = (uint8_t)(row[i] + pred);
            }
        }
        // ---- to RGBA8 ----
        for (uint32_t y = 0; y < ph; y++) {
            const uint8_t *row = base + (size_t)y * (stride + 1) + 1;
            for (uint32_t x = 0; x < pw; x++) {
                uint32_t s[4] = {0, 0, 0, 0}, s16[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < chans; k++) {
                    if (depth == 8)
                        s[k] = row[(size_t)x * chans + k];
                    else if (depth == 16) {
                        s[k] = row[((size_t)x * chans + k) * 2]; // the high byte, as lodepng's 16 -> 8 conversion
                        s16[k] = ((uint32_t)s[k] << 8) | row[((size_t)x * chans + k) * 2 + 1];
                    } else {
                        const size_t bit = (size_t)x * depth;
                        s[k] = (row[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
                    }
                }
                uint8_t *px = &rgba[((size_t)(ps.y0 + y * ps.dy) * w + (ps.x0 + x * ps.dx)) * 4];
                px[3] = 255;
                if (ctype == 3) {
                    if (plte.size() < (size_t)s[0] * 3 + 3) return err = "palette index out of range", false;
                    px[0] = plte[s[0] * 3], px[1] = plte[s[0] * 3 + 1], px[2] = plte[s[0] * 3 + 2];
                    if (s[0] < trns.size()) px[3] = trns[s[0]];
                } else if (ctype == 0 || ctype == 4) {
                    const uint8_t g = depth < 8 ? (uint8_t)(s[0] * 255 / maxv) : (uint8_t)s[0];
                    px[0] = px[1] = px[2] = g;
                    if (ctype == 4) px[3] = (uint8_t)s[1];
                    if (ctype == 0 && trns.size() >= 2) {
                        const uint32_t key = ((uint32_t)trns[0] << 8) | trns[1];
                        if ((depth == 16 ? s16[0] : s[0]) == key) px[3] = 0;
                    }
                } else {
                    px[0] = (uint8_t)s[0], px[1] = (uint8_t)s[1], px[2] = (uint8_t)s[2];
                    if (ctype == 6) px[3] = (uint8_t)s[3];
                    if (ctype == 2 && trns.size() >= 6) {
                        bool eq = true;
                        for (int k = 0; k < 3; k++) eq = eq && (depth == 16 ? s16[k] : s[k]) == (((uint32_t)trns[2 * k] << 8) | trns[2 * k + 1]);
                        if (eq) px[3] = 0;
                    }
                }
            }
        }
    }
    return true;
}

inline bool read_file(const char *name, std::vector<uint8_t> &data)
{
    FILE *f = fopen(name, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    data.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && fread(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    return ok;
}

} // namespace png_loader
