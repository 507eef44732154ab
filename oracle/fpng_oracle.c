/*
 * fpng_oracle.c -- CPU restatement of the fpng encode hot path.  TEST INFRASTRUCTURE ONLY
 * (see fpng_oracle.h).  Parity status: PINNED against oracle/_ref (the unmodified reference) and
 * tests/golden/.
 *
 * The restatement is deliberately written in the ROW-LOCAL form the GPU kernels use, not in the
 * reference's serial-bit-buffer form:
 *   - the filtered row is a pure function of raw rows y and y-1            (src/fpng.cpp:1592-1660, :1696)
 *   - a row's token bit string depends only on that row's filtered bytes   (src/fpng.cpp:1468-1558)
 *   - an RLE chunk is emitted by its LAST pixel (look-back only + 1 px look-ahead), which is
 *     equivalent to the reference's greedy forward scan from the run start (src/fpng.cpp:1503-1514)
 *   - the "encode failed, use stored blocks" outcome is evaluated in closed form from the final
 *     bit position and the size of the last flush unit (src/fpng.cpp:567-588) instead of by
 *     running out of buffer.
 * All citations are into /root/reference/.
 */
#include "fpng_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Format constants.  The two trained 1-pass Deflate block prefixes ARE the file format
 * (src/fpng.cpp:532-535 and :548-551): zlib header 78 01, BFINAL=1, BTYPE=2 and the complete
 * dynamic-Huffman code-length header.  The {code,size} tables the reference also stores
 * (src/fpng.cpp:536-546, :552-562) are NOT duplicated here: they are re-derived by parsing the
 * prefix the way any inflater would, which doubles as a self-check.
 * ------------------------------------------------------------------------------------------ */
static const uint8_t k_prefix3[62] = {
    0x78, 0x01, 0xED, 0xC3, 0x03, 0xB0, 0x6E, 0x59, 0x7A, 0x80, 0xE1, 0xF7, 0xFB, 0xD6, 0xDA, 0xF8,
    0x71, 0x7C, 0xAD, 0xBE, 0x6D, 0x0C, 0x32, 0xC9, 0xC4, 0xB6, 0x6D, 0xDB, 0xB6, 0x6D, 0xDB, 0xB6,
    0x6D, 0xDB, 0xC9, 0x24, 0x93, 0x99, 0x69, 0xEB, 0xF6, 0x35, 0x8E, 0xCF, 0x8F, 0x8D, 0xB5, 0xD6,
    0x97, 0x5D, 0x75, 0xAA, 0x4E, 0x75, 0x75, 0x3A, 0xCE, 0x4D, 0xD2, 0xD9, 0xA9, 0x7A};
static const uint8_t k_prefix4[61] = {
    0x78, 0x01, 0xE5, 0xC4, 0x63, 0xB4, 0x25, 0x67, 0xDA, 0x80, 0xE1, 0xFB, 0x79, 0xAB, 0x6A, 0xF3,
    0xD8, 0xE7, 0xB4, 0x6D, 0xC4, 0xB6, 0x33, 0x33, 0x49, 0x06, 0xC9, 0xD8, 0xB6, 0x6D, 0xDB, 0xB6,
    0x11, 0x8C, 0x62, 0xDB, 0x66, 0xDB, 0x3C, 0x7D, 0xAC, 0xCD, 0xAA, 0x7A, 0x9F, 0x6F, 0xD5, 0x8F,
    0xB3, 0xD6, 0x5E, 0xBD, 0x3A, 0x99, 0x68, 0xA6, 0x67, 0xBE, 0xF7, 0xC7, 0x75};
/* Bits already pending after the whole prefix bytes (src/fpng.cpp:535, :551). */
#define PREFIX3_TAIL_BITS 7u
#define PREFIX3_TAIL_VAL 30u
#define PREFIX4_TAIL_BITS 2u
#define PREFIX4_TAIL_VAL 1u

/* Order in which code-length-code lengths are stored (RFC 1951 3.2.7; src/fpng.cpp:728). */
static const uint8_t k_clc_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

typedef struct {
    uint8_t len[288];
    uint16_t code[288]; /* already bit-reversed for LSB-first emission */
} huff_table;

static huff_table g_tab3, g_tab4;
static uint16_t g_len_sym[256]; /* match_len-3 -> length symbol   (src/fpng.cpp:498-507) */
static uint8_t g_len_extra[256]; /* match_len-3 -> # extra bits    (src/fpng.cpp:509-512) */
static uint32_t g_crc_tab[8][256];
static int g_ready = 0;

static uint32_t bitrev(uint32_t v, unsigned n)
{
    uint32_t r = 0;
    for (unsigned i = 0; i < n; i++, v >>= 1) r = (r << 1) | (v & 1u);
    return r;
}

/* Canonical code assignment from code lengths (RFC 1951 3.2.2), stored bit-reversed
 * (src/fpng.cpp:699-708). */
static void canonical_codes(const uint8_t *len, unsigned n, unsigned max_len, uint16_t *code)
{
    unsigned count[17] = {0}, next[17] = {0};
    for (unsigned i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    unsigned c = 0;
    for (unsigned l = 1; l <= max_len; l++) {
        c = (c + count[l - 1]) << 1;
        next[l] = c;
    }
    for (unsigned i = 0; i < n; i++) {
        code[i] = 0;
        if (len[i]) code[i] = (uint16_t)bitrev(next[len[i]]++, len[i]);
    }
}

/* LSB-first bit reader over `bytes` followed by a few pending tail bits. */
typedef struct {
    const uint8_t *p;
    uint32_t nbytes, tail_val, tail_bits, pos;
} bitr;
static uint32_t br_bit(bitr *r)
{
    uint32_t byte = r->pos >> 3, b;
    if (byte < r->nbytes)
        b = (r->p[byte] >> (r->pos & 7)) & 1u;
    else
        b = (r->tail_val >> (r->pos - 8 * r->nbytes)) & 1u;
    r->pos++;
    return b;
}
static uint32_t br_bits(bitr *r, unsigned n)
{
    uint32_t v = 0;
    for (unsigned i = 0; i < n; i++) v |= br_bit(r) << i;
    return v;
}
/* Decode one symbol of a canonical code given as (len, bit-reversed code) pairs. */
static int br_sym(bitr *r, const uint8_t *len, const uint16_t *code, unsigned n)
{
    uint32_t acc = 0;
    for (unsigned l = 1; l <= 15; l++) {
        acc |= br_bit(r) << (l - 1);
        for (unsigned s = 0; s < n; s++)
            if (len[s] == l && code[s] == acc) return (int)s;
    }
    return -1;
}

/* Parse a dynamic-block header (RFC 1951 3.2.7) -> literal/length code lengths.
 * Returns the bit position after the header, or 0 on a malformed header. */
static uint32_t parse_prefix(const uint8_t *bytes, uint32_t nbytes, uint32_t tail_val, uint32_t tail_bits,
                             huff_table *t)
{
    bitr r = {bytes, nbytes, tail_val, tail_bits, 16}; /* skip zlib CMF/FLG */
    if (bytes[0] != 0x78 || bytes[1] != 0x01) return 0;
    if (br_bits(&r, 1) != 1) return 0; /* BFINAL */
    if (br_bits(&r, 2) != 2) return 0; /* BTYPE = dynamic */
    unsigned hlit = br_bits(&r, 5) + 257, hdist = br_bits(&r, 5) + 1, hclen = br_bits(&r, 4) + 4;
    uint8_t cl_len[19] = {0};
    uint16_t cl_code[19];
    for (unsigned i = 0; i < hclen; i++) cl_len[k_clc_order[i]] = (uint8_t)br_bits(&r, 3);
    canonical_codes(cl_len, 19, 7, cl_code);
    uint8_t lens[288 + 32];
    memset(lens, 0, sizeof lens);
    unsigned i = 0;
    while (i < hlit + hdist) {
        int s = br_sym(&r, cl_len, cl_code, 19);
        if (s < 0) return 0;
        if (s < 16) {
            lens[i++] = (uint8_t)s;
        } else {
            unsigned rep, val = 0;
            if (s == 16) {
                if (!i) return 0;
                val = lens[i - 1];
                rep = 3 + br_bits(&r, 2);
            } else if (s == 17)
                rep = 3 + br_bits(&r, 3);
            else
                rep = 11 + br_bits(&r, 7);
            if (i + rep > hlit + hdist) return 0;
            while (rep--) lens[i++] = (uint8_t)val;
        }
    }
    memset(t->len, 0, sizeof t->len);
    memcpy(t->len, lens, hlit);
    canonical_codes(t->len, 288, 15, t->code);
    return r.pos;
}

static void init_crc_tables(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        g_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int k = 1; k < 8; k++) g_crc_tab[k][i] = (g_crc_tab[k - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[k - 1][i] & 0xFF];
}

static void init_len_tables(void)
{
    /* RFC 1951 3.2.5 length code table, re-indexed by match_len-3. */
    static const uint16_t base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                      31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (unsigned len = 3; len <= 258; len++) {
        unsigned s = 28;
        while (base[s] > len) s--;
        g_len_sym[len - 3] = (uint16_t)(257 + s);
        g_len_extra[len - 3] = extra[s];
    }
}

static void ensure_init(void)
{
    if (g_ready) return;
    init_crc_tables();
    init_len_tables();
    uint32_t e3 = parse_prefix(k_prefix3, sizeof k_prefix3, PREFIX3_TAIL_VAL, PREFIX3_TAIL_BITS, &g_tab3);
    uint32_t e4 = parse_prefix(k_prefix4, sizeof k_prefix4, PREFIX4_TAIL_VAL, PREFIX4_TAIL_BITS, &g_tab4);
    /* the header must end exactly where the reference starts appending tokens */
    if (e3 != 62 * 8 + PREFIX3_TAIL_BITS || e4 != 61 * 8 + PREFIX4_TAIL_BITS) abort();
    g_ready = 1;
}

void fpo_get_1pass_table(uint32_t num_chans, uint8_t len_out[288], uint16_t code_out[288], const uint8_t **prefix,
                         uint32_t *prefix_len, uint32_t *start_bit)
{
    ensure_init();
    const huff_table *t = (num_chans == 3) ? &g_tab3 : &g_tab4;
    memcpy(len_out, t->len, 288);
    memcpy(code_out, t->code, 288 * sizeof(uint16_t));
    if (prefix) *prefix = (num_chans == 3) ? k_prefix3 : k_prefix4;
    if (prefix_len) *prefix_len = (num_chans == 3) ? 62 : 61;
    if (start_bit) *start_bit = (num_chans == 3) ? 62 * 8 + PREFIX3_TAIL_BITS : 61 * 8 + PREFIX4_TAIL_BITS;
}

void fpo_get_len_tables(uint16_t len_sym[256], uint8_t len_extra[256])
{
    ensure_init();
    memcpy(len_sym, g_len_sym, sizeof g_len_sym);
    memcpy(len_extra, g_len_extra, sizeof g_len_extra);
}

/* ------------------------------------------------------------------------------------------
 * Checksums
 * ------------------------------------------------------------------------------------------ */

/* CRC-32/ISO-HDLC, same calling convention as fpng_crc32 (src/fpng.cpp:234-253, :393-401):
 * prev_crc is the finished CRC of the preceding bytes (0 for none). */
uint32_t fpo_crc32(const void *data, size_t size, uint32_t prev_crc)
{
    ensure_init();
    const uint8_t *p = (const uint8_t *)data;
    uint32_t c = ~prev_crc;
    while (size >= 8) {
        uint32_t a = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        a ^= c;
        c = g_crc_tab[7][a & 0xFF] ^ g_crc_tab[6][(a >> 8) & 0xFF] ^ g_crc_tab[5][(a >> 16) & 0xFF] ^
            g_crc_tab[4][a >> 24] ^ g_crc_tab[3][p[4]] ^ g_crc_tab[2][p[5]] ^ g_crc_tab[1][p[6]] ^ g_crc_tab[0][p[7]];
        p += 8;
        size -= 8;
    }
    while (size--) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 0xFF];
    return ~c;
}

/* Adler-32 (RFC 1950), same calling convention as fpng_adler32 (src/fpng.cpp:465-487). */
uint32_t fpo_adler32(const void *data, size_t size, uint32_t adler)
{
    const uint8_t *p = (const uint8_t *)data;
    uint32_t s1 = adler & 0xFFFF, s2 = adler >> 16;
    while (size) {
        size_t n = size < 5552 ? size : 5552; /* largest n with no u32 overflow before the modulo */
        size -= n;
        while (n--) {
            s1 += *p++;
            s2 += s1;
        }
        s1 %= 65521u;
        s2 %= 65521u;
    }
    return (s2 << 16) | s1;
}

/* ------------------------------------------------------------------------------------------
 * Bit string writer (LSB-first, src/fpng.cpp:564-565 semantics without the flush bookkeeping)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint8_t *buf; /* zero-initialised */
    size_t cap;
    uint64_t pos; /* bit position */
    int overflow;
} bitw;

static void bw_put(bitw *b, uint32_t v, unsigned n)
{
    if (!n) return;
    if (((b->pos + n + 7) >> 3) > b->cap) {
        b->overflow = 1;
        b->pos += n;
        return;
    }
    uint64_t x = (uint64_t)v << (b->pos & 7);
    size_t i = (size_t)(b->pos >> 3);
    for (; x; x >>= 8, i++) b->buf[i] |= (uint8_t)x;
    b->pos += n;
}

/* ------------------------------------------------------------------------------------------
 * Row walker: the token grammar of one row range.
 *
 * mode HIST : accumulate lit_freq[] exactly as pass 1 of the 2-pass coders
 *             (src/fpng.cpp:1021-1084 / :1299-1363)
 * mode EMIT : append token bits with table t
 *             (1-pass: src/fpng.cpp:1182-1241 / :1468-1558; 2-pass pass 2: :1112-1143 / :1385-1429)
 * lit_test  : the 4-channel 1-pass "are 4 literals cheaper than a 1-pixel match" rule
 *             (src/fpng.cpp:1520-1528) -- never active for 3 channels or 2-pass.
 * Also accumulates the Adler raw sums of the filtered bytes and reports the size in bits of the
 * last flush unit (needed by the failure rule, see encode()).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *img;
    uint32_t w, h, c;
    const huff_table *t;
    bitw *bw;
    uint32_t *hist;
    int lit_test;
    int filter_none; /* raw fallback: every row filter 0 */
    int one_pass_3ch_units; /* 1-pass 3ch: filter literal and px0 share one flush unit */
    uint32_t last_unit_bits;
    uint64_t s1, s2, nbytes; /* Adler raw sums: s1 = sum d_i, s2 = sum (n-i) d_i, both mod 65521 */
} walker;

static unsigned lit_bits(const walker *k, const uint8_t *px)
{
    unsigned n = 0;
    for (uint32_t i = 0; i < k->c; i++) n += k->t->len[px[i]];
    return n;
}

static void put_lit(walker *k, unsigned sym)
{
    if (k->hist) k->hist[sym]++;
    if (k->bw) bw_put(k->bw, k->t->code[sym], k->t->len[sym]);
}

static unsigned put_pixel_lits(walker *k, const uint8_t *px)
{
    unsigned n = 0;
    for (uint32_t i = 0; i < k->c; i++) {
        put_lit(k, px[i]);
        if (k->t) n += k->t->len[px[i]];
    }
    return n;
}

/* One RLE chunk of n pixels = one match of n*c bytes at distance c
 * (src/fpng.cpp:1221-1226 / :1515-1533): length symbol, then extra bits + the 1-bit distance
 * code which is always 0. */
static unsigned put_chunk(walker *k, unsigned n_px)
{
    unsigned adj = n_px * k->c - 3, sym = g_len_sym[adj], ex = g_len_extra[adj], bits = 0;
    if (k->hist) k->hist[sym]++;
    if (k->t) bits = k->t->len[sym] + ex + 1;
    if (k->bw) {
        bw_put(k->bw, k->t->code[sym], k->t->len[sym]);
        bw_put(k->bw, adj & ((1u << ex) - 1u), ex + 1);
    }
    return bits;
}

static void walk_rows(walker *k, uint32_t y0, uint32_t y1)
{
    const uint32_t c = k->c, w = k->w, bpl = w * c;
    const unsigned cap = (c == 4) ? 63 : 85; /* 252/4, 255/3  (src/fpng.cpp:1506, :1212) */
    uint8_t *f = (uint8_t *)malloc((size_t)bpl + 1);
    for (uint32_t y = y0; y < y1; y++) {
        const uint8_t *cur = k->img + (size_t)y * bpl;
        /* filter: None on row 0, Up elsewhere (src/fpng.cpp:1696, :1598-1655) */
        if (y == 0 || k->filter_none) {
            f[0] = 0;
            memcpy(f + 1, cur, bpl);
        } else {
            const uint8_t *up = cur - bpl;
            f[0] = 2;
            for (uint32_t i = 0; i < bpl; i++) f[1 + i] = (uint8_t)(cur[i] - up[i]);
        }
        /* Adler raw sums of this row appended to what we have */
        {
            uint64_t a = 0, b = 0, n = (uint64_t)bpl + 1;
            for (uint64_t i = 0; i < n; i++) {
                a += f[i];
                b += (n - i) * f[i];
                if ((i & 0xFFF) == 0xFFF) b %= 65521u;
            }
            a %= 65521u;
            b %= 65521u;
            k->s2 = (k->s2 + (n % 65521u) * k->s1 + b) % 65521u;
            k->s1 = (k->s1 + a) % 65521u;
            k->nbytes += n;
        }
        if (!k->bw && !k->hist) continue;

        const uint8_t *P = f + 1;
        put_lit(k, f[0]);
        unsigned unit = k->t ? k->t->len[f[0]] : 0;
        unsigned px0 = put_pixel_lits(k, P);
        k->last_unit_bits = k->one_pass_3ch_units ? unit + px0 : px0;
        unsigned q = 0; /* pixels in the currently open chunk */
        for (uint32_t x = 1; x < w; x++) {
            const uint8_t *px = P + (size_t)x * c;
            if (memcmp(px, px - c, c) == 0) {
                q++;
                int next_same = (x + 1 < w) && memcmp(px + c, px, c) == 0;
                if (q == cap || !next_same) {
                    unsigned adj = q * c - 3;
                    if (k->lit_test && q == 1 && (unsigned)(k->t->len[g_len_sym[adj]] + g_len_extra[adj] + 1) > lit_bits(k, px))
                        k->last_unit_bits = put_pixel_lits(k, px);
                    else
                        k->last_unit_bits = put_chunk(k, q);
                    q = 0;
                }
            } else {
                k->last_unit_bits = put_pixel_lits(k, px);
            }
        }
    }
    free(f);
}

/* ------------------------------------------------------------------------------------------
 * 2-pass table construction
 * ------------------------------------------------------------------------------------------ */

/* Code lengths for `n` symbols with 16-bit counts, limited to max_len, plus canonical codes.
 * Restates defl_optimize_huffman_table (src/fpng.cpp:676-709): stable sort of the used symbols by
 * count (radix sort there, :622-637 -- any stable sort gives the same order), minimum-redundancy
 * code lengths with the two-queue construction that ties towards the LEAF (:639-661, weights kept
 * in 16 bits like the in-place original), Kraft repair (:663-674), shortest codes handed to the
 * END of the sorted order (:697-698), canonical bit-reversed codes (:699-708). */
static void build_table(const uint16_t *count, unsigned n, unsigned max_len, uint8_t *len, uint16_t *code)
{
    uint16_t key[288], sym[288];
    unsigned used = 0;
    for (unsigned i = 0; i < n; i++)
        if (count[i]) {
            key[used] = count[i];
            sym[used++] = (uint16_t)i;
        }
    /* stable insertion sort, ascending by key */
    for (unsigned i = 1; i < used; i++) {
        uint16_t k0 = key[i], s0 = sym[i];
        unsigned j = i;
        for (; j > 0 && key[j - 1] > k0; j--) {
            key[j] = key[j - 1];
            sym[j] = sym[j - 1];
        }
        key[j] = k0;
        sym[j] = s0;
    }
    int num_codes[40];
    memset(num_codes, 0, sizeof num_codes);
    if (used == 1) {
        num_codes[1] = 1;
    } else if (used >= 2) {
        /* two-queue Huffman: leaves [0,used) ascending, internal nodes in creation order */
        uint16_t iw[288];
        int iparent[288], lparent[288];
        unsigned leaf = 0, root = 0, made = 0;
        while (made < used - 1) {
            unsigned pick[2];
            int is_leaf[2];
            uint16_t wsum = 0;
            for (int k = 0; k < 2; k++) {
                /* take the internal node only if it is strictly lighter than the next leaf */
                if (leaf >= used || (root < made && iw[root] < key[leaf])) {
                    is_leaf[k] = 0;
                    pick[k] = root++;
                    wsum = (uint16_t)(wsum + iw[pick[k]]);
                } else {
                    is_leaf[k] = 1;
                    pick[k] = leaf++;
                    wsum = (uint16_t)(wsum + key[pick[k]]);
                }
            }
            for (int k = 0; k < 2; k++) {
                if (is_leaf[k])
                    lparent[pick[k]] = (int)made;
                else
                    iparent[pick[k]] = (int)made;
            }
            iw[made++] = wsum;
        }
        int idepth[288];
        idepth[made - 1] = 0;
        for (int j = (int)made - 2; j >= 0; j--) idepth[j] = idepth[iparent[j]] + 1;
        for (unsigned i = 0; i < used; i++) num_codes[idepth[lparent[i]] + 1]++;
        /* Kraft repair to max_len */
        for (unsigned l = max_len + 1; l < 40; l++) {
            num_codes[max_len] += num_codes[l];
            num_codes[l] = 0;
        }
        uint32_t total = 0;
        for (unsigned l = max_len; l > 0; l--) total += (uint32_t)num_codes[l] << (max_len - l);
        while (total != (1u << max_len)) {
            num_codes[max_len]--;
            for (unsigned l = max_len - 1; l > 0; l--)
                if (num_codes[l]) {
                    num_codes[l]--;
                    num_codes[l + 1] += 2;
                    break;
                }
            total--;
        }
    }
    memset(len, 0, n);
    unsigned j = used;
    for (unsigned l = 1; l <= max_len; l++)
        for (int r = num_codes[l]; r > 0; r--) len[sym[--j]] = (uint8_t)l;
    canonical_codes(len, n, max_len, code);
}

/* adjust_freq32 (src/fpng.cpp:868-907): scale to 16 bits, never to zero.  The reference's
 * "total > 65535" repair loop only rewrites the 32-bit input array, which nobody reads
 * afterwards, so it has no observable effect and is not restated. */
static void adjust_freq(const uint32_t freq[288], uint16_t c0[288])
{
    uint64_t total = 0;
    for (unsigned i = 0; i < 288; i++) total += freq[i];
    total &= 0xFFFFFFFFu; /* total_freq is uint32_t there */
    for (unsigned i = 0; i < 288; i++) {
        if (!freq[i] || !total) {
            c0[i] = 0;
            continue;
        }
        uint32_t s = (uint32_t)(((uint64_t)freq[i] * 65535u) / total);
        c0[i] = (uint16_t)(s ? s : 1);
    }
}

static uint32_t build_dynamic_from_counts(uint16_t c0[288], uint32_t num_chans, uint8_t len_out[288], uint16_t code_out[288], uint8_t *hdr);

uint32_t fpo_build_dynamic_table(const uint32_t lit_freq_in[288], uint32_t num_chans, uint8_t len_out[288],
                                 uint16_t code_out[288], uint8_t *hdr)
{
    ensure_init();
    uint32_t freq[288];
    memcpy(freq, lit_freq_in, sizeof freq);
    freq[256] = 1; /* src/fpng.cpp:1092 / :1371 */
    uint16_t c0[288];
    adjust_freq(freq, c0);
    return build_dynamic_from_counts(c0, num_chans, len_out, code_out, hdr);
}

/* Table training (src/fpng_test.cpp:766-973 training_mode + src/fpng.cpp:909-988 create_dynamic_block_prefix, both under
 * FPNG_TRAIN_HUFFMAN_TABLES): every image of the corpus (all with num_chans channels) is encoded 2-pass; what is summed is
 * its 16-bit ADJUSTED histogram as it enters defl_start_dynamic_block (src/fpng.cpp:751-755: before the end-of-block count is
 * forced to 1).  The sums (truncated to 32 bits, :932) get every literal, the end-of-block symbol and every length symbol a
 * multiple-of-num_chans match can use set to at least 1 (:941-952), are adjusted to 16 bits again (:954) and go through the
 * same builder and header writer as a 2-pass image.  Returns the header length in bits like fpo_build_dynamic_table. */
uint32_t fpo_train_tables(const void *const *images, const uint32_t *w, const uint32_t *h, uint32_t n, uint32_t num_chans,
                          uint8_t len_out[288], uint16_t code_out[288], uint8_t *hdr)
{
    ensure_init();
    uint64_t sum[288];
    memset(sum, 0, sizeof sum);
    for (uint32_t k = 0; k < n; k++) {
        uint32_t freq[288];
        uint16_t c0[288];
        fpo_band_hist(images[k], w[k], h[k], num_chans, 0, h[k], freq);
        freq[256] = 1; /* src/fpng.cpp:1092 / :1371 */
        adjust_freq(freq, c0);
        for (unsigned i = 0; i < 288; i++) sum[i] += c0[i];
    }
    uint32_t lit_freq[288];
    for (unsigned i = 0; i < 288; i++) lit_freq[i] = (uint32_t)sum[i];
    for (unsigned i = 0; i <= 256; i++)
        if (!lit_freq[i]) lit_freq[i] = 1;
    for (uint32_t len = num_chans; len <= 258; len += num_chans)
        if (!lit_freq[g_len_sym[len - 3]]) lit_freq[g_len_sym[len - 3]] = 1;
    uint16_t c0[288];
    adjust_freq(lit_freq, c0);
    return build_dynamic_from_counts(c0, num_chans, len_out, code_out, hdr);
}

static uint32_t build_dynamic_from_counts(uint16_t c0[288], uint32_t num_chans, uint8_t len_out[288], uint16_t code_out[288], uint8_t *hdr)
{
    uint16_t c1[32], c2[19];
    c0[256] = 1; /* src/fpng.cpp:757 */
    memset(c1, 0, sizeof c1);
    c1[num_chans - 1] = 1; /* distance symbol of distance 3 / 4 (src/fpng.cpp:1019, :1097-1098) */
    c1[num_chans] = 1;     /* dummy neighbour ("wuffs workaround") */

    huff_table lit;
    uint8_t dlen[32];
    uint16_t dcode[32];
    build_table(c0, 288, 12, lit.len, lit.code);
    build_table(c1, 32, 12, dlen, dcode);

    unsigned n_lit = 286, n_dist = 30;
    while (n_lit > 257 && !lit.len[n_lit - 1]) n_lit--;
    while (n_dist > 1 && !dlen[n_dist - 1]) n_dist--;
    uint8_t seq[288 + 32];
    memcpy(seq, lit.len, n_lit);
    memcpy(seq + n_lit, dlen, n_dist);
    unsigned n_seq = n_lit + n_dist;

    /* run-length packing of the code lengths (src/fpng.cpp:711-726, :770-794) */
    uint8_t packed[2 * (288 + 32)];
    unsigned n_packed = 0, zrun = 0, rep = 0, prev = 0xFF;
    memset(c2, 0, sizeof c2);
#define FLUSH_REP()                                                   \
    do {                                                              \
        if (rep) {                                                    \
            if (rep < 3) {                                            \
                c2[prev] = (uint16_t)(c2[prev] + rep);                \
                while (rep--) packed[n_packed++] = (uint8_t)prev;     \
            } else {                                                  \
                c2[16]++;                                             \
                packed[n_packed++] = 16;                              \
                packed[n_packed++] = (uint8_t)(rep - 3);              \
            }                                                         \
            rep = 0;                                                  \
        }                                                             \
    } while (0)
#define FLUSH_ZERO()                                                  \
    do {                                                              \
        if (zrun) {                                                   \
            if (zrun < 3) {                                           \
                c2[0] = (uint16_t)(c2[0] + zrun);                     \
                while (zrun--) packed[n_packed++] = 0;                \
            } else if (zrun <= 10) {                                  \
                c2[17]++;                                             \
                packed[n_packed++] = 17;                              \
                packed[n_packed++] = (uint8_t)(zrun - 3);             \
            } else {                                                  \
                c2[18]++;                                             \
                packed[n_packed++] = 18;                              \
                packed[n_packed++] = (uint8_t)(zrun - 11);            \
            }                                                         \
            zrun = 0;                                                 \
        }                                                             \
    } while (0)
    for (unsigned i = 0; i < n_seq; i++) {
        unsigned cs = seq[i];
        if (!cs) {
            FLUSH_REP();
            if (++zrun == 138) FLUSH_ZERO();
        } else {
            FLUSH_ZERO();
            if (cs != prev) {
                FLUSH_REP();
                c2[cs]++;
                packed[n_packed++] = (uint8_t)cs;
            } else if (++rep == 6) {
                FLUSH_REP();
            }
        }
        prev = cs;
    }
    if (rep)
        FLUSH_REP();
    else
        FLUSH_ZERO();
#undef FLUSH_REP
#undef FLUSH_ZERO

    uint8_t cl_len[19];
    uint16_t cl_code[19];
    build_table(c2, 19, 7, cl_len, cl_code);

    memset(hdr, 0, 400);
    bitw b = {hdr, 400, 0, 0};
    bw_put(&b, 0x78, 8); /* src/fpng.cpp:1279-1283 */
    bw_put(&b, 0x01, 8);
    bw_put(&b, 1, 1);    /* BFINAL */
    bw_put(&b, 2, 2);    /* BTYPE  (src/fpng.cpp:799) */
    bw_put(&b, n_lit - 257, 5);
    bw_put(&b, n_dist - 1, 5);
    int nbl = 18;
    while (nbl >= 0 && !cl_len[k_clc_order[nbl]]) nbl--;
    nbl = (nbl + 1 < 4) ? 4 : nbl + 1;
    bw_put(&b, (uint32_t)nbl - 4, 4);
    for (int i = 0; i < nbl; i++) bw_put(&b, cl_len[k_clc_order[i]], 3);
    for (unsigned i = 0; i < n_packed;) {
        unsigned s = packed[i++];
        bw_put(&b, cl_code[s], cl_len[s]);
        if (s >= 16) bw_put(&b, packed[i++], s == 16 ? 2 : (s == 17 ? 3 : 7));
    }
    memcpy(len_out, lit.len, 288);
    memcpy(code_out, lit.code, sizeof lit.code);
    return (uint32_t)b.pos;
}

/* ------------------------------------------------------------------------------------------
 * Whole-image encode
 * ------------------------------------------------------------------------------------------ */
size_t fpo_max_encoded_size(uint32_t w, uint32_t h, uint32_t num_chans)
{
    uint64_t n = ((uint64_t)w * num_chans + 1) * h;
    return (size_t)(58 + 6 + n + 5 * ((n + 65534) / 65535) + 16);
}

static void put_be32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

/* Stored-block stream of the filter-0 image (src/fpng.cpp:818-866, :1728-1758). */
static size_t write_stored(const uint8_t *img, uint32_t w, uint32_t h, uint32_t c, uint8_t *dst)
{
    const uint64_t bpl = (uint64_t)w * c, n = (bpl + 1) * h;
    uint8_t *s = (uint8_t *)malloc((size_t)n);
    for (uint32_t y = 0; y < h; y++) {
        s[(size_t)(y * (bpl + 1))] = 0;
        memcpy(s + (size_t)(y * (bpl + 1)) + 1, img + (size_t)(y * bpl), (size_t)bpl);
    }
    size_t o = 0;
    dst[o++] = 0x78;
    dst[o++] = 0x01;
    for (uint64_t i = 0; i < n;) {
        uint32_t blk = (uint32_t)((n - i < 65535) ? (n - i) : 65535);
        dst[o++] = (i + blk == n) ? 1 : 0;
        dst[o++] = (uint8_t)blk;
        dst[o++] = (uint8_t)(blk >> 8);
        dst[o++] = (uint8_t)~blk;
        dst[o++] = (uint8_t)(~blk >> 8);
        memcpy(dst + o, s + i, blk);
        o += blk;
        i += blk;
    }
    put_be32(dst + o, fpo_adler32(s, (size_t)n, 1));
    free(s);
    return o + 4;
}

int fpo_encode(const void *image, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t flags, uint8_t *out,
               size_t out_cap, size_t *out_size)
{
    ensure_init();
    /* argument rules: src/fpng.cpp:1670-1680 */
    if (w < 1 || h < 1 || (uint64_t)w * h > 0xFFFFFFFFull || w > (1u << 24) || h > (1u << 24)) return 0;
    if (num_chans != 3 && num_chans != 4) return 0;
    const uint64_t n = ((uint64_t)w * num_chans + 1) * h;
    if (n > 0xFFFFFF00ull) return 0; /* reference arithmetic is 32-bit here (src/fpng.cpp:1682-1705): undefined there */
    if (out_cap < fpo_max_encoded_size(w, h, num_chans)) return 0;

    /* byte budget handed to the coder: src/fpng.cpp:1705, :1713-1722 */
    const uint64_t D = ((58 + n + 7) & ~7ull) - 58;
    uint8_t *z = out + 58;
    size_t zlen = 0;

    if (!(flags & FPO_FORCE_UNCOMPRESSED)) {
        const int two_pass = (flags & FPO_ENCODE_SLOWER) != 0;
        /* worst case: every pixel literal at 12 bits per byte, plus one filter literal per row */
        size_t scratch_cap = (size_t)((((uint64_t)w * num_chans + 1) * h * 12 + 7) / 8) + 512;
        uint8_t *scratch = (uint8_t *)calloc(scratch_cap, 1);
        bitw bw = {scratch, scratch_cap, 0, 0};
        walker k;
        memset(&k, 0, sizeof k);
        k.img = (const uint8_t *)image;
        k.w = w;
        k.h = h;
        k.c = num_chans;
        huff_table dyn;
        unsigned eob_len;
        int ok = 1;
        if (!two_pass) {
            const uint8_t *pre = (num_chans == 3) ? k_prefix3 : k_prefix4;
            uint32_t pre_len = (num_chans == 3) ? 62 : 61;
            if (D < pre_len) ok = 0; /* src/fpng.cpp:1169, :1455 */
            memcpy(scratch, pre, pre_len);
            bw.pos = (uint64_t)pre_len * 8;
            if (num_chans == 3)
                bw_put(&bw, PREFIX3_TAIL_VAL, PREFIX3_TAIL_BITS);
            else
                bw_put(&bw, PREFIX4_TAIL_VAL, PREFIX4_TAIL_BITS);
            k.t = (num_chans == 3) ? &g_tab3 : &g_tab4;
            k.lit_test = (num_chans == 4);
            k.one_pass_3ch_units = (num_chans == 3);
        } else {
            uint32_t hist[288];
            memset(hist, 0, sizeof hist);
            walker hk = k;
            hk.hist = hist;
            walk_rows(&hk, 0, h);
            uint8_t hdr[400];
            uint32_t hbits = fpo_build_dynamic_table(hist, num_chans, dyn.len, dyn.code, hdr);
            memcpy(scratch, hdr, (hbits + 7) / 8);
            bw.pos = hbits;
            k.t = &dyn;
        }
        eob_len = k.t->len[256];
        k.bw = &bw;
        walk_rows(&k, 0, h);
        const uint64_t s_last = bw.pos;
        /* Failure rule, closed form of the buffer-slack checks in PUT_BITS_FLUSH
         * (src/fpng.cpp:567-576): every flush needs 8 writable bytes at the byte offset reached by
         * the PREVIOUS flush; offsets are monotone so only the final flush matters. */
        const uint64_t s_pen = s_last - k.last_unit_bits;
        if ((s_pen >> 3) + 8 > D) ok = 0;
        /* EOB, byte alignment and the 4 Adler bytes must fit (src/fpng.cpp:578-588, :1564-1577) */
        if (((s_last + eob_len + 7) >> 3) + 4 > D) ok = 0;
        if (ok) {
            bw_put(&bw, k.t->code[256], eob_len);
            zlen = (size_t)((bw.pos + 7) >> 3);
            memcpy(z, scratch, zlen);
            uint32_t s1 = (uint32_t)((1 + k.s1) % 65521u), s2 = (uint32_t)((k.nbytes % 65521u + k.s2) % 65521u);
            put_be32(z + zlen, (s2 << 16) | s1);
            zlen += 4;
        }
        free(scratch);
    }
    if (!zlen) zlen = write_stored((const uint8_t *)image, w, h, num_chans, z);

    /* container: src/fpng.cpp:1764-1800 */
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A};
    static const uint8_t fdec[17] = {0, 0, 0, 5, 'f', 'd', 'E', 'C', 82, 36, 147, 227, 0, 0xE5, 0xAB, 0x62, 0x99};
    static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
    memcpy(out, sig, 8);
    put_be32(out + 8, 13);
    memcpy(out + 12, "IHDR", 4);
    /* only the low 16 bits of each dimension are stored (src/fpng.cpp:1773-1774) -- reproduced */
    put_be32(out + 16, w & 0xFFFF);
    put_be32(out + 20, h & 0xFFFF);
    out[24] = 8;
    out[25] = (num_chans == 3) ? 2 : 6;
    out[26] = out[27] = out[28] = 0;
    put_be32(out + 29, fpo_crc32(out + 12, 17, 0));
    memcpy(out + 33, fdec, 17);
    put_be32(out + 50, (uint32_t)zlen);
    memcpy(out + 54, "IDAT", 4);
    put_be32(out + 58 + zlen, fpo_crc32(out + 54, zlen + 4, 0));
    memcpy(out + 58 + zlen + 4, iend, 12);
    *out_size = 58 + zlen + 16;
    return 1;
}

uint64_t fpo_encode_band_1pass(const void *image, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t y0, uint32_t y1,
                               uint8_t *out, size_t out_cap, uint32_t *adler_s1, uint32_t *adler_s2,
                               uint64_t *adler_len, uint32_t *last_unit_bits)
{
    ensure_init();
    bitw bw = {out, out_cap, 0, 0};
    walker k;
    memset(&k, 0, sizeof k);
    k.img = (const uint8_t *)image;
    k.w = w;
    k.h = h;
    k.c = num_chans;
    k.t = (num_chans == 3) ? &g_tab3 : &g_tab4;
    k.lit_test = (num_chans == 4);
    k.one_pass_3ch_units = (num_chans == 3);
    k.bw = &bw;
    walk_rows(&k, y0, y1);
    if (adler_s1) *adler_s1 = (uint32_t)k.s1;
    if (adler_s2) *adler_s2 = (uint32_t)k.s2;
    if (adler_len) *adler_len = k.nbytes;
    if (last_unit_bits) *last_unit_bits = k.last_unit_bits;
    return bw.overflow ? 0 : bw.pos;
}

/* Row bands with either table: the symbol histogram of rows [y0,y1) (2-pass, pass 1; the histograms of an image's
 * bands add up to the whole image's, src/fpng.cpp:1021-1084 / :1299-1363), and the band's token bits under the 1-pass
 * table (lit_freq == NULL) or under the table built from the WHOLE image's histogram.  *first_token_bit = where row
 * tokens start in the zlib stream, hdr (>= 400 bytes) receives the stream's head (zlib header + block header). */
void fpo_band_hist(const void *image, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t y0, uint32_t y1, uint32_t hist[288])
{
    ensure_init();
    walker k;
    memset(&k, 0, sizeof k);
    memset(hist, 0, 288 * sizeof(uint32_t));
    k.img = (const uint8_t *)image;
    k.w = w, k.h = h, k.c = num_chans;
    k.hist = hist;
    walk_rows(&k, y0, y1);
}

uint64_t fpo_encode_band(const void *image, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t y0, uint32_t y1,
                         const uint32_t *lit_freq, uint8_t *out, size_t out_cap, uint32_t *adler_s1, uint32_t *adler_s2,
                         uint64_t *adler_len, uint32_t *last_unit_bits, uint32_t *first_token_bit, uint32_t *eob_bits,
                         uint32_t *eob_code, uint8_t *hdr)
{
    ensure_init();
    bitw bw = {out, out_cap, 0, 0};
    walker k;
    huff_table dyn;
    memset(&k, 0, sizeof k);
    k.img = (const uint8_t *)image;
    k.w = w, k.h = h, k.c = num_chans;
    if (lit_freq) {
        *first_token_bit = fpo_build_dynamic_table(lit_freq, num_chans, dyn.len, dyn.code, hdr);
        k.t = &dyn;
    } else {
        const uint8_t *pre = (num_chans == 3) ? k_prefix3 : k_prefix4;
        const uint32_t pre_len = (num_chans == 3) ? 62 : 61;
        memset(hdr, 0, 400);
        memcpy(hdr, pre, pre_len);
        hdr[pre_len] = (num_chans == 3) ? PREFIX3_TAIL_VAL : PREFIX4_TAIL_VAL;
        *first_token_bit = pre_len * 8 + ((num_chans == 3) ? PREFIX3_TAIL_BITS : PREFIX4_TAIL_BITS);
        k.t = (num_chans == 3) ? &g_tab3 : &g_tab4;
        k.lit_test = (num_chans == 4);
        k.one_pass_3ch_units = (num_chans == 3);
    }
    *eob_bits = k.t->len[256];
    *eob_code = k.t->code[256];
    k.bw = &bw;
    walk_rows(&k, y0, y1);
    if (adler_s1) *adler_s1 = (uint32_t)k.s1;
    if (adler_s2) *adler_s2 = (uint32_t)k.s2;
    if (adler_len) *adler_len = k.nbytes;
    if (last_unit_bits) *last_unit_bits = k.last_unit_bits;
    return bw.overflow ? 0 : bw.pos;
}
