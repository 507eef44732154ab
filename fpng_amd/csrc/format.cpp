// format.cpp -- host-side derivation of the fpng format tables and host checksum utilities.
#include "format.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "kernels.h"

#include <array>
#include <cstring>
#include <mutex>
#include <vector>

namespace fpng_amd {

// The trained single-pass block prefixes (reference src/fpng.cpp:532-535 for RGB, :548-551 for
// RGBA).  They contain the zlib header, BFINAL/BTYPE and the whole dynamic code-length header;
// the bits pending after the last whole byte are kept separately, as in the reference.
static const uint8_t kPrefixRGB[62] = {
    0x78, 0x01, 0xED, 0xC3, 0x03, 0xB0, 0x6E, 0x59, 0x7A, 0x80, 0xE1, 0xF7, 0xFB, 0xD6, 0xDA, 0xF8,
    0x71, 0x7C, 0xAD, 0xBE, 0x6D, 0x0C, 0x32, 0xC9, 0xC4, 0xB6, 0x6D, 0xDB, 0xB6, 0x6D, 0xDB, 0xB6,
    0x6D, 0xDB, 0xC9, 0x24, 0x93, 0x99, 0x69, 0xEB, 0xF6, 0x35, 0x8E, 0xCF, 0x8F, 0x8D, 0xB5, 0xD6,
    0x97, 0x5D, 0x75, 0xAA, 0x4E, 0x75, 0x75, 0x3A, 0xCE, 0x4D, 0xD2, 0xD9, 0xA9, 0x7A};
static const uint8_t kPrefixRGBA[61] = {
    0x78, 0x01, 0xE5, 0xC4, 0x63, 0xB4, 0x25, 0x67, 0xDA, 0x80, 0xE1, 0xFB, 0x79, 0xAB, 0x6A, 0xF3,
    0xD8, 0xE7, 0xB4, 0x6D, 0xC4, 0xB6, 0x33, 0x33, 0x49, 0x06, 0xC9, 0xD8, 0xB6, 0x6D, 0xDB, 0xB6,
    0x11, 0x8C, 0x62, 0xDB, 0x66, 0xDB, 0x3C, 0x7D, 0xAC, 0xCD, 0xAA, 0x7A, 0x9F, 0x6F, 0xD5, 0x8F,
    0xB3, 0xD6, 0x5E, 0xBD, 0x3A, 0x99, 0x68, 0xA6, 0x67, 0xBE, 0xF7, 0xC7, 0x75};
struct PendingBits {
    uint32_t value, count;
};
static const PendingBits kTailRGB = {30, 7};  // reference src/fpng.cpp:535
static const PendingBits kTailRGBA = {1, 2};  // reference src/fpng.cpp:551

void deflate_length_symbol(uint32_t adj_len, uint32_t *sym, uint32_t *extra_bits)
{
    // RFC 1951 3.2.5: symbols 257..264 cover lengths 3..10 one each; then groups of four symbols
    // share e = 1..5 extra bits; length 258 is symbol 285 with no extra bits.
    const uint32_t len = adj_len + 3;
    if (len == 258) {
        *sym = 285;
        *extra_bits = 0;
        return;
    }
    if (len <= 10) {
        *sym = 254 + len;
        *extra_bits = 0;
        return;
    }
    uint32_t e = 1, base = 11, s = 265;
    while (len >= base + (4u << e)) {
        base += 4u << e;
        s += 4;
        e++;
    }
    *sym = s + ((len - base) >> e);
    *extra_bits = e;
}

namespace {

// Minimal LSB-first bit cursor over the prefix bytes + pending bits.
class PrefixBits {
public:
    PrefixBits(const uint8_t *p, uint32_t n, PendingBits tail) : bits_()
    {
        bits_.reserve(n * 8 + tail.count);
        for (uint32_t i = 0; i < n; i++)
            for (int b = 0; b < 8; b++) bits_.push_back((p[i] >> b) & 1);
        for (uint32_t b = 0; b < tail.count; b++) bits_.push_back((tail.value >> b) & 1);
    }
    bool has(uint32_t n) const { return pos_ + n <= bits_.size(); }
    uint32_t take(uint32_t n)
    {
        uint32_t v = 0;
        for (uint32_t i = 0; i < n; i++) v |= (uint32_t)bits_[pos_++] << i;
        return v;
    }
    uint32_t pos() const { return pos_; }
    uint32_t size() const { return (uint32_t)bits_.size(); }

private:
    std::vector<uint8_t> bits_;
    uint32_t pos_ = 0;
};

// Canonical Huffman codes (RFC 1951 3.2.2) from lengths; returned MSB-first.
template <size_t N> std::array<uint16_t, N> canonical_msb_first(const std::array<uint8_t, N> &len)
{
    uint32_t per_len[16] = {0};
    for (uint8_t l : len) per_len[l]++;
    per_len[0] = 0;
    uint32_t first[16] = {0}, code = 0;
    for (int l = 1; l < 16; l++) {
        code = (code + per_len[l - 1]) << 1;
        first[l] = code;
    }
    std::array<uint16_t, N> out{};
    for (size_t s = 0; s < N; s++)
        if (len[s]) out[s] = (uint16_t)first[len[s]]++;
    return out;
}

uint32_t reverse_bits(uint32_t v, uint32_t n)
{
    uint32_t r = 0;
    while (n--) {
        r = (r << 1) | (v & 1);
        v >>= 1;
    }
    return r;
}

bool parse_prefix(const uint8_t *prefix, uint32_t nbytes, PendingBits tail, uint32_t num_chans, TokenTable *t)
{
    PrefixBits in(prefix, nbytes, tail);
    if (in.take(8) != 0x78 || in.take(8) != 0x01) return false; // zlib CMF/FLG
    if (in.take(1) != 1 || in.take(2) != 2) return false;      // BFINAL, BTYPE=dynamic
    const uint32_t n_lit = in.take(5) + 257, n_dist = in.take(5) + 1, n_clc = in.take(4) + 4;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    std::array<uint8_t, 19> clc_len{};
    for (uint32_t i = 0; i < n_clc; i++) clc_len[order[i]] = (uint8_t)in.take(3);
    const auto clc_code = canonical_msb_first(clc_len);

    std::vector<uint8_t> lens;
    while (lens.size() < n_lit + n_dist) {
        // walk the code-length code one bit at a time (MSB-first comparison)
        uint32_t acc = 0, sym = 0xFFFF;
        for (uint32_t l = 1; l <= 7 && sym == 0xFFFF; l++) {
            if (!in.has(1)) return false;
            acc = (acc << 1) | in.take(1);
            for (uint32_t s = 0; s < 19; s++)
                if (clc_len[s] == l && clc_code[s] == acc) sym = s;
        }
        if (sym == 0xFFFF) return false;
        if (sym < 16)
            lens.push_back((uint8_t)sym);
        else {
            uint32_t rep = (sym == 16) ? 3 + in.take(2) : (sym == 17) ? 3 + in.take(3) : 11 + in.take(7);
            uint8_t v = 0;
            if (sym == 16) {
                if (lens.empty()) return false;
                v = lens.back();
            }
            lens.insert(lens.end(), rep, v);
        }
    }
    if (lens.size() != n_lit + n_dist || in.pos() != in.size()) return false; // header must end exactly at the tail

    std::array<uint8_t, 288> lit_len{};
    for (uint32_t s = 0; s < n_lit; s++) lit_len[s] = lens[s];
    // the distance tree must give the used distance (3 or 4) the 1-bit code 0 (reference src/fpng.cpp:1099 assert)
    if (lens[n_lit + num_chans - 1] != 1) return false;
    const auto lit_code_msb = canonical_msb_first(lit_len);

    std::memset(t, 0, sizeof(*t));
    for (uint32_t s = 0; s < 288; s++)
        t->lit[s] = reverse_bits(lit_code_msb[s], lit_len[s]) | ((uint32_t)lit_len[s] << 16);
    const uint32_t cap = (num_chans == 3) ? kMaxChunkPixels3 : kMaxChunkPixels4;
    for (uint32_t q = 1; q <= cap; q++) {
        uint32_t adj = q * num_chans - 3, sym, extra;
        deflate_length_symbol(adj, &sym, &extra);
        if (!lit_len[sym]) return false; // the trained table must be able to code every chunk length
        // extra-bits value = offset within the symbol's length group = low `extra` bits of adj
        const uint32_t code = lit_code(t->lit[sym]) | ((adj & ((1u << extra) - 1u)) << lit_len[sym]);
        t->chunk[q] = code | ((lit_len[sym] + extra + 1u) << 24);
    }
    t->first_token_bit = t->header_bits = in.size();
    std::memcpy(t->header, prefix, nbytes);
    t->header[nbytes] = (uint8_t)tail.value;
    return true;
}

uint32_t g_crc_byte[256];
uint32_t g_crc_slice[16][256]; // g_crc_slice[k][b]: CRC state after byte b followed by k zero bytes (k = 0: g_crc_byte)
std::once_flag g_crc_once;
void ensure_crc() // callable from any thread (fpng_crc32 is a free function of the re-entrant drop-in)
{
    std::call_once(g_crc_once, [] {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
            g_crc_byte[i] = c;
            g_crc_slice[0][i] = c;
        }
        for (int k = 1; k < 16; k++)
            for (uint32_t i = 0; i < 256; i++) g_crc_slice[k][i] = (g_crc_slice[k - 1][i] >> 8) ^ g_crc_byte[g_crc_slice[k - 1][i] & 0xFF];
    });
}

} // namespace

bool build_1pass_tables(TokenTable *t3, TokenTable *t4)
{
    return parse_prefix(kPrefixRGB, sizeof kPrefixRGB, kTailRGB, 3, t3) &&
           parse_prefix(kPrefixRGBA, sizeof kPrefixRGBA, kTailRGBA, 4, t4) && t3->first_token_bit == 503 &&
           t4->first_token_bit == 490;
}

#if defined(__x86_64__)
// CRC-32 by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction",
// Intel 2009; the constants are x^k mod P for the reflected polynomial, as every zlib-compatible implementation of the method
// uses them): four 128-bit lanes folded 512 bits ahead per step, then 128 bits at a time, then a Barrett reduction.  `c` is the
// running state (already inverted), len >= 64 and a multiple of 16.  The reference has its own version of the same method
// (src/fpng.cpp:255-292); results are checked against zlib's crc32 in tests/test_abi.py.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(const uint8_t *p, size_t len, uint32_t c)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
#define ld(q) _mm_loadu_si128((const __m128i *)(q))
    __m128i x1 = _mm_xor_si128(ld(p), _mm_cvtsi32_si128((int)c)), x2 = ld(p + 16), x3 = ld(p + 32), x4 = ld(p + 48);
    p += 64, len -= 64;
    while (len >= 64) {
        const __m128i l1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), l2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i l3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), l4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k1k2, 0x11), l1), ld(p));
        x2 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x2, k1k2, 0x11), l2), ld(p + 16));
        x3 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x3, k1k2, 0x11), l3), ld(p + 32));
        x4 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x4, k1k2, 0x11), l4), ld(p + 48));
        p += 64, len -= 64;
    }
#define FPNG_FOLD128(acc, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(acc, k3k4, 0x11), _mm_clmulepi64_si128(acc, k3k4, 0x00)), next)
    x1 = FPNG_FOLD128(x1, x2);
    x1 = FPNG_FOLD128(x1, x3);
    x1 = FPNG_FOLD128(x1, x4);
    for (; len >= 16; p += 16, len -= 16) x1 = FPNG_FOLD128(x1, _mm_loadu_si128((const __m128i *)p));
#undef FPNG_FOLD128
    // 128 -> 64 -> 32 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), t);
    t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10), mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x00);
    return (uint32_t)_mm_extract_epi32(_mm_xor_si128(x1, t), 1);
#undef ld
}
static bool have_clmul()
{
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return ok;
}
#endif

uint32_t host_crc32(const void *data, size_t size, uint32_t prev)
{
    ensure_crc();
    const uint8_t *p = static_cast<const uint8_t *>(data);
    uint32_t c = ~prev;
#if defined(__x86_64__)
    if (size >= 64 && have_clmul()) {
        const size_t body = size & ~(size_t)15;
        c = crc32_clmul(p, body, c);
        p += body, size -= body;
    }
#endif
    // sixteen bytes per step (slicing-by-16: every byte's contribution to the state 16 bytes later is a table entry, the sixteen
    // lookups are independent); what the reference does with carry-less multiplies (src/fpng.cpp:255-292) or four bytes at a time
    size_t i = 0;
    for (; i + 16 <= size; i += 16) {
        uint32_t w0, w1, w2, w3;
        std::memcpy(&w0, p + i, 4), std::memcpy(&w1, p + i + 4, 4), std::memcpy(&w2, p + i + 8, 4), std::memcpy(&w3, p + i + 12, 4);
        w0 ^= c;
        c = g_crc_slice[15][w0 & 0xFF] ^ g_crc_slice[14][(w0 >> 8) & 0xFF] ^ g_crc_slice[13][(w0 >> 16) & 0xFF] ^ g_crc_slice[12][w0 >> 24] ^
            g_crc_slice[11][w1 & 0xFF] ^ g_crc_slice[10][(w1 >> 8) & 0xFF] ^ g_crc_slice[9][(w1 >> 16) & 0xFF] ^ g_crc_slice[8][w1 >> 24] ^
            g_crc_slice[7][w2 & 0xFF] ^ g_crc_slice[6][(w2 >> 8) & 0xFF] ^ g_crc_slice[5][(w2 >> 16) & 0xFF] ^ g_crc_slice[4][w2 >> 24] ^
            g_crc_slice[3][w3 & 0xFF] ^ g_crc_slice[2][(w3 >> 8) & 0xFF] ^ g_crc_slice[1][(w3 >> 16) & 0xFF] ^ g_crc_slice[0][w3 >> 24];
    }
    for (; i < size; i++) c = (c >> 8) ^ g_crc_byte[(c ^ p[i]) & 0xFF];
    return ~c;
}

uint32_t host_adler32(const void *data, size_t size, uint32_t prev)
{
    const uint8_t *p = static_cast<const uint8_t *>(data);
    uint32_t a = prev & 0xFFFF, b = prev >> 16;
    size_t i = 0;
    while (i < size) {
        // 5552 bytes at most between reductions (the largest n with 255 n (n + 1) / 2 + (n + 1) 65520 < 2^32); inside, sixteen
        // bytes per step: b gains 16 a + sum of (16 - j) p[j] -- independent multiply-adds instead of a chain of 32 additions
        size_t end = (size - i > 5552) ? i + 5552 : size;
#if defined(__x86_64__)
        {
            // (SSE2, the x86-64 baseline) byte sums with psadbw, weighted sums with pmaddwd, three vector accumulators per
            // chunk: vs = byte sums so far, vps = sum of vs in front of every block (b gains 16 of them per block), vw = weighted
            const __m128i zero = _mm_setzero_si128();
            const __m128i w_lo = _mm_setr_epi16(16, 15, 14, 13, 12, 11, 10, 9), w_hi = _mm_setr_epi16(8, 7, 6, 5, 4, 3, 2, 1);
            __m128i vs = zero, vps = zero, vw = zero;
            size_t nblk = 0;
            for (; i + 16 <= end; i += 16, nblk++) {
                const __m128i v = _mm_loadu_si128((const __m128i *)(p + i));
                vps = _mm_add_epi32(vps, vs);
                vs = _mm_add_epi32(vs, _mm_sad_epu8(v, zero));
                vw = _mm_add_epi32(vw, _mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi8(v, zero), w_lo), _mm_madd_epi16(_mm_unpackhi_epi8(v, zero), w_hi)));
            }
            auto hsum = [](__m128i x) {
                alignas(16) uint32_t t[4];
                _mm_store_si128((__m128i *)t, x);
                return (uint64_t)t[0] + t[1] + t[2] + t[3];
            };
            const uint64_t b64 = (uint64_t)b + 16ull * nblk * a + 16ull * hsum(vps) + hsum(vw);
            a = (uint32_t)(((uint64_t)a + hsum(vs)) % kAdlerMod);
            b = (uint32_t)(b64 % kAdlerMod);
        }
#endif
        for (; i + 16 <= end; i += 16) {
            uint32_t s = 0, ws = 0;
            for (uint32_t j = 0; j < 16; j++) s += p[i + j], ws += (16u - j) * p[i + j];
            b += 16u * a + ws;
            a += s;
        }
        for (; i < end; i++) {
            a += p[i];
            b += a;
        }
        a %= kAdlerMod;
        b %= kAdlerMod;
    }
    return (b << 16) | a;
}

// Reflected-domain polynomial product modulo the CRC-32 polynomial: bit 31 of a word is x^0.
uint32_t gf2_mulmod(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
    for (int i = 31; i >= 0 && a; i--) {
        if (a & (1u << i)) {
            r ^= b;
            a &= ~(1u << i);
        }
        b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; // multiply b by x
    }
    return r;
}

uint32_t gf2_xpow8n(uint64_t nbytes)
{
    // square-and-multiply on x^8 ; x^8 in the reflected domain is bit (31-8)
    uint32_t result = 0x80000000u, base = 0x00800000u;
    while (nbytes) {
        if (nbytes & 1) result = gf2_mulmod(result, base);
        base = gf2_mulmod(base, base);
        nbytes >>= 1;
    }
    return result;
}

uint32_t crc32_combine(uint32_t crc_x, uint32_t crc_y, uint64_t len_y)
{
    return gf2_mulmod(crc_x, gf2_xpow8n(len_y)) ^ crc_y;
}

uint32_t adler32_combine(uint32_t adler_x, uint32_t adler_y, uint64_t len_y)
{
    const uint64_t a1 = adler_x & 0xFFFF, b1 = adler_x >> 16, a2 = adler_y & 0xFFFF, b2 = adler_y >> 16;
    const uint64_t n = len_y % kAdlerMod;
    // X contributes (a1-1) to every running sum while Y is consumed
    const uint64_t a = (a1 + a2 + kAdlerMod - 1) % kAdlerMod;
    const uint64_t b = (b1 + b2 + n * ((a1 + kAdlerMod - 1) % kAdlerMod)) % kAdlerMod;
    return (uint32_t)((b << 16) | a);
}

uint32_t gf2_xpow(uint64_t e)
{
    uint32_t result = 0x80000000u, base = 0x40000000u; // 1, x
    while (e) {
        if (e & 1) result = gf2_mulmod(result, base);
        base = gf2_mulmod(base, base);
        e >>= 1;
    }
    return result;
}

void build_crc_device_tables(CrcDeviceTables *t)
{
    ensure_crc();
    std::memset(t, 0, sizeof(*t));
    // raw (init 0, no final xor) CRC of a single byte b followed by z zero bytes = crc_byte[b] * x^(8z)
    for (int k = 0; k < 16; k++) {
        const uint32_t shift = gf2_xpow8n(15 - k + kCrcRowBytes - 16);
        for (uint32_t b = 0; b < 256; b++) t->striped[k][b] = gf2_mulmod(g_crc_byte[b], shift);
    }
    for (uint32_t i = 0; i < 256; i++) t->lane_fix[i] = gf2_xpow8n(kCrcRowBytes - 16 * i);
    for (uint32_t i = 0; i < 48; i++) t->pow2[i] = gf2_xpow8n(1ull << i);
    const uint64_t ord = 0xFFFFFFFFull; // multiplicative order of x divides 2^32-1 (P is irreducible)
    for (uint32_t p = 0; p < 16; p++) t->inv_pad[p] = gf2_xpow((ord - 8ull * p) % ord);
    t->inv_row = gf2_xpow((ord - 8ull * kCrcRowBytes % ord) % ord);
    for (uint32_t p = 0; p < 16; p++) t->inv_row_pad[p] = gf2_mulmod(t->inv_row, t->inv_pad[p]);
    for (uint32_t e = 12; e <= 24; e++)
        for (uint64_t i = 0; i < 256; i++) t->fold[e - 12][i] = gf2_xpow8n(i << e);
    for (uint32_t k = 0; k < 6; k++)
        for (uint64_t b = 0; b < 256; b++) t->pow_byte[k][b] = gf2_xpow8n(b << (8 * k));
}

} // namespace fpng_amd
