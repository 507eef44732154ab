// decode_core.h -- the per-thread logic of the GPU decoder (decode.hip), written so that it also compiles for the host:
// tests/cpp/decode_emul.cpp runs the same walkers thread by thread on the CPU (no GPU in the dev container), and
// tools/sync_stats.c measured the numbers the constants come from.  Reference: src/fpng.cpp:2209-2901 decodes the same streams
// serially, one token at a time; here a stream is cut into subsequences of kSubBits token bits that are decoded independently.
//
// THE LOOKUP TABLE (built on the host, decode_api.cpp: build_multi_lut).  Index = the next 12 stream bits, one 32-bit entry:
//   bits 31..28  L: code bits this entry consumes (1..12); 0 = no such code
//   bits 27..26  n: number of LITERALS decoded at once (1..3: as many whole literal codes as fit into the 12 bits, at most 3);
//                   their byte values in bits 7..0, 15..8, 23..16 (first one lowest)
//   n == 0, bit 25 set:   a match length symbol: bits 8..0 base length (3..258), bits 11..9 number of extra bits (0..5); L covers
//                         the symbol's code only (the extra bits and the 1-bit distance code follow in the stream)
//   n == 0, bit 25 clear: end of block
// Behind the 4096 entries: lenof[256], the code length of every literal byte value -- a group of literals is taken apart with it
// where token granularity matters (the hand-over between two subsequences must not depend on how the literals were grouped).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FPNG_DEC_HD __host__ __device__ __forceinline__
#else
#define FPNG_DEC_HD inline
#endif

namespace fpng_amd {
namespace dec {

constexpr uint32_t kLutEntries = 4096;
constexpr uint32_t kLutDwords = kLutEntries + 64; // + lenof[256]
constexpr uint32_t kEntMatch = 1u << 25;
enum : uint32_t { kSubEob = 1u, kSubInvalid = 4u };
enum : uint32_t { kTokLit = 0, kTokMatch = 1, kTokEob = 2, kTokInvalid = 3 };

// (hi:lo) >> sh, 0 <= sh <= 31
FPNG_DEC_HD uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}

// One lookup.  w: the stream bits from the token's first bit on (at least 18 valid bits); room > 0: how many bits lie between
// the token's first bit and the bit at which the caller stops -- a group of literals is cut down to the tokens that START in
// front of that bit.  Out: n literal bytes in `lits` (kTokLit), `run` bytes (kTokMatch); bits consumed.
FPNG_DEC_HD uint32_t fetch(uint32_t w, const uint32_t *lut, const uint8_t *lenof, uint32_t room, uint32_t &n, uint32_t &lits, uint32_t &run, uint32_t &bits)
{
    const uint32_t e = lut[w & (kLutEntries - 1)];
    const uint32_t L = e >> 28;
    n = (e >> 26) & 3u;
    bits = L;
    if (!L) return kTokInvalid;
    if (n) {
        lits = e & 0xFFFFFFu;
        if (L > room) { // the group reaches over the stopping bit: token by token
            uint32_t k = 0, used = 0;
            do {
                used += lenof[(lits >> (8 * k)) & 255u];
                k++;
            } while (k < n && used < room);
            n = k, bits = used;
            lits &= 0xFFFFFFu >> (8 * (3 - k));
        }
        return kTokLit;
    }
    if (e & kEntMatch) {
        const uint32_t xb = (e >> 9) & 7u;
        run = (e & 511u) + ((w >> L) & ((1u << xb) - 1u));
        bits = L + xb + 1; // extra bits + the 1-bit distance code ("the previous pixel": reference src/fpng.cpp:2301)
        return kTokMatch;
    }
    return kTokEob;
}

// what a subsequence's decode leaves behind
struct SubCount {
    uint32_t bytes; // output bytes of its tokens
    uint32_t lits;  // ... of which literals
    uint32_t tail;  // its last four literal bytes (the most recent one in bits 31..24)
    uint32_t flags; // kSubEob: it met an end-of-block symbol; kSubInvalid: its decode derailed
};

// Decodes the tokens that start in [pos, limit) (positions: bits relative to the staged slice); returns the position behind the
// last one.  A 32-bit window serves two lookups when the first one used at most 14 bits (12 + 5 extra + 1 + 14 <= 32).
template <bool Count, class Bits>
FPNG_DEC_HD uint32_t walk_count(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t pos, uint32_t limit, uint32_t data_limit, SubCount &c)
{
    while (pos < limit) {
        if (pos >= data_limit) { // ran off the data without an end-of-block symbol
            c.flags = kSubInvalid;
            break;
        }
        const uint32_t w = in.window(pos);
        uint32_t used = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint32_t n, lits = 0, run = 0, bits;
            const uint32_t kind = fetch(w >> used, lut, lenof, limit - (pos + used), n, lits, run, bits);
            if (kind >= kTokEob) {
                c.flags = kind == kTokEob ? kSubEob : kSubInvalid;
                return pos + used;
            }
            used += bits;
            if (Count) {
                if (kind == kTokLit) {
                    c.bytes += n, c.lits += n;
                    c.tail = funnel(lits, c.tail, 8 * n);
                } else
                    c.bytes += run;
            }
            if (used > 14 || pos + used >= limit) break;
        }
        pos += used;
    }
    return pos;
}

// a subsequence as the synchronisation keeps it
struct SubState {
    uint32_t start, end; // first bit of its first token; position behind its last one (its nominal boundary if it is flagged)
    SubCount c;
};

// First decode of a subsequence [nominal, boundary): the decoder starts `lead` bits EARLIER and has, with a probability that
// tools/sync_stats.c measured (128 bits: all but 0.04 % of the subsequences of a synthetic gradient, 1.8 % of a photograph),
// fallen into step with the true token sequence when it crosses `nominal`; the first token boundary at or behind `nominal` is
// the subsequence's start.  Whether it is the true one shows when it is compared with the predecessor's end.
template <class Bits>
FPNG_DEC_HD void sub_first(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t lead_start, uint32_t nominal, uint32_t boundary, uint32_t data_limit, SubState &s)
{
    uint32_t p = nominal;
    if (lead_start < nominal) {
        SubCount d = {0, 0, 0, 0};
        p = walk_count<false>(in, lut, lenof, lead_start, nominal, data_limit, d);
        if (d.flags || p < nominal) p = nominal; // the lead-in derailed: any start is as good as another
    }
    s.start = p;
    s.c.bytes = s.c.lits = s.c.tail = s.c.flags = 0;
    const uint32_t e = walk_count<true>(in, lut, lenof, p, boundary, data_limit, s.c);
    // A decode that derailed or met an end-of-block symbol hands over on the nominal boundary: speculative decodes meet FALSE
    // end-of-block symbols; which one is the true one is settled afterwards (the first one of the chain).
    s.end = s.c.flags ? boundary : e;
}

// The subsequence must start at `want` instead of s.start (its predecessor ended there).  Both decodes -- the old one from
// s.start, the new one from `want` -- are stepped token by token, the one that lags behind first; where they meet, the rest of the
// old decode holds, and only the counts in front of that point are exchanged.  No meeting point inside the subsequence (or one
// so late that the last four literals are not all behind it): decoded again as a whole.
template <class Bits>
FPNG_DEC_HD void sub_refix(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t want, uint32_t boundary, uint32_t data_limit, SubState &s)
{
    uint32_t A = s.start, B = want;
    SubCount a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    while (A != B) {
        if ((A < B ? A : B) >= boundary) break;
        if (A < B) {
            A = walk_count<true>(in, lut, lenof, A, A + 1, data_limit, a);
            if (a.flags) break;
        } else {
            B = walk_count<true>(in, lut, lenof, B, B + 1, data_limit, b);
            if (b.flags) break;
        }
    }
    s.start = want;
    if (A == B && !a.flags && !b.flags && s.c.lits - a.lits >= 4) {
        s.c.bytes = s.c.bytes - a.bytes + b.bytes;
        s.c.lits = s.c.lits - a.lits + b.lits;
        return; // (end, tail and flags are the old decode's)
    }
    s.c.bytes = s.c.lits = s.c.tail = s.c.flags = 0;
    const uint32_t e = walk_count<true>(in, lut, lenof, want, boundary, data_limit, s.c);
    s.end = s.c.flags ? boundary : e;
}

// per-subsequence record in global memory: start - nominal (0..17) | end - boundary (0..17) << 5 | flags << 10 | literals << 13
FPNG_DEC_HD uint32_t pack_info(uint32_t start_rel, uint32_t end_rel, const SubCount &c) { return start_rel | end_rel << 5 | c.flags << 10 | c.lits << 13; }
FPNG_DEC_HD uint32_t info_start(uint32_t v) { return v & 31u; }
FPNG_DEC_HD uint32_t info_end(uint32_t v) { return (v >> 5) & 31u; }
FPNG_DEC_HD uint32_t info_flags(uint32_t v) { return (v >> 10) & 7u; }
FPNG_DEC_HD uint32_t info_lits(uint32_t v) { return v >> 13; }

// ---- the real decode: one subsequence's tokens into the workgroup's tile of the filtered stream ----
// The filtered stream = what the reference's decoder consumes row by row (src/fpng.cpp:2255-2262): h rows of 1 filter byte +
// w * c bytes, kept in exactly this layout (the column kernels read it with unaligned loads).  A tile is a window of kTileBytes
// stream bytes held in LDS; Tile::put32(d, v) / put8(b, v) store into it (indices relative to the window, already checked
// against it by the walker: a subsequence that straddles two tiles is decoded by both workgroups, each keeps its part).
struct EmitGeom {
    uint32_t stride; // w * c + 1
    uint32_t c;      // channels in the file
    uint32_t ndw;    // dwords of the tile window
};
enum : uint32_t { kEmitBadStream = 2u, kEmitSawEob = 0x100u };

template <class Tile> struct TileWriter {
    Tile &tile;
    uint32_t ndw;
    uint64_t acc;  // bytes not stored yet, the oldest one lowest
    uint32_t have; // how many (0..3 between tokens)
    int32_t dw;    // tile dword they go to
    uint32_t skip; // leading bytes of that dword that belong to the previous subsequence (first dword only)
    FPNG_DEC_HD TileWriter(Tile &t, uint32_t ndw_, int32_t rel) : tile(t), ndw(ndw_), acc(0)
    {
        const int32_t al = rel & ~3;
        have = skip = (uint32_t)(rel - al);
        dw = al >> 2;
    }
    FPNG_DEC_HD void flush() // have >= 4
    {
        if ((uint32_t)dw < ndw) {
            if (!skip)
                tile.put32((uint32_t)dw, (uint32_t)acc);
            else
                for (uint32_t b = skip; b < 4; b++) tile.put8((uint32_t)dw * 4 + b, (uint8_t)(acc >> (8 * b)));
        }
        skip = 0, acc >>= 32, have -= 4, dw++;
    }
    FPNG_DEC_HD void put(uint32_t bytes, uint32_t n) // n <= 4 bytes, the first one lowest
    {
        acc |= (uint64_t)bytes << (8 * have);
        have += n;
        if (have >= 4) flush();
    }
    FPNG_DEC_HD void finish()
    {
        if ((uint32_t)dw < ndw)
            for (uint32_t b = skip; b < have; b++) tile.put8((uint32_t)dw * 4 + b, (uint8_t)(acc >> (8 * b)));
    }
    // npix copies of a 4-byte pixel: the first dword completes the pending bytes, the others are one rotated constant
    FPNG_DEC_HD void run4(uint32_t px, uint32_t npix)
    {
        const uint64_t x = (uint64_t)px << (8 * have);
        acc |= x; // (x's low `have` bytes are zero)
        const uint32_t keep = have;
        have += 4;
        flush();
        const uint32_t r = (uint32_t)x | (uint32_t)(x >> 32);
        const int32_t j1 = dw + (int32_t)npix - 1;
        int32_t j = dw < 0 ? 0 : dw;
        const int32_t stop = j1 < (int32_t)ndw ? j1 : (int32_t)ndw;
        for (; j < stop; j++) tile.put32((uint32_t)j, r);
        dw = j1, acc = x >> 32, have = keep;
    }
    // npix copies of a 3-byte pixel; runs that lie wholly in front of or behind the window only move the position
    FPNG_DEC_HD void run3(uint32_t px, uint32_t npix)
    {
        const uint32_t total = have + 3 * npix;
        if (total >= 8 && (dw >= (int32_t)ndw || dw + (int32_t)(total >> 2) < 0)) {
            const uint32_t m = total & 3u; // the last m bytes of the pixel stay pending
            dw += (int32_t)(total >> 2), have = m, skip = 0;
            acc = m ? (px >> (8 * (3 - m))) : 0u;
            return;
        }
        for (uint32_t k = 0; k < npix; k++) put(px, 3);
    }
};

// pos / limit / data_limit as in walk_count; rel = tile-relative stream byte of the subsequence's first output byte (may be
// negative), (row, col) = its place in the image (col 0 = the filter byte), lastpx = the four literal bytes in front of it.
// Returns kEmit* flags; eob_end = position behind the end-of-block symbol if it met one.
template <class Bits, class Tile>
FPNG_DEC_HD uint32_t walk_emit(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t pos, uint32_t limit, uint32_t data_limit, int32_t rel, uint32_t row,
                               uint32_t col, uint32_t lastpx, const EmitGeom &g, Tile &tile, uint32_t &eob_end)
{
    TileWriter<Tile> out(tile, g.ndw, rel);
    if (col == 1) lastpx = 0; // right behind a filter byte there is no previous pixel: zeros (reference :2262 prev_delta_* = 0)
    uint32_t err = 0;
    const uint32_t stride = g.stride, c = g.c;
    bool stop = false;
    while (pos < limit && !stop) {
        if (pos >= data_limit) {
            err = kEmitBadStream;
            break;
        }
        const uint32_t w = in.window(pos);
        uint32_t used = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint32_t n, lits = 0, run = 0, bits;
            const uint32_t kind = fetch(w >> used, lut, lenof, limit - (pos + used), n, lits, run, bits);
            used += bits;
            if (kind == kTokLit) {
                if (col && col + n <= stride) { // inside the row
                    out.put(lits, n);
                    lastpx = funnel(lits, lastpx, 8 * n);
                    col += n;
                    if (col == stride) col = 0, row++;
                } else {
                    for (uint32_t j = 0; j < n; j++) {
                        const uint32_t b = (lits >> (8 * j)) & 255u;
                        out.put(b, 1);
                        if (!col) { // the row's filter literal: 0, then 2 = Up (reference :2255-2259)
                            if (b != (row ? 2u : 0u)) err = kEmitBadStream;
                            lastpx = 0, col = 1;
                        } else {
                            lastpx = funnel(b, lastpx, 8);
                            if (++col == stride) col = 0, row++;
                        }
                    }
                }
            } else if (kind == kTokMatch) {
                // a match repeats the previous pixel: whole pixels, starting on a pixel, inside the row (reference :2273-2330)
                const uint32_t x = col - 1;
                const bool whole = c == 4 ? !((x | run) & 3u) : (x % 3u == 0 && run % 3u == 0);
                if (!col || !whole || col + run > stride) {
                    err = kEmitBadStream;
                    stop = true;
                    break;
                }
                if (c == 4)
                    out.run4(lastpx, run >> 2);
                else
                    out.run3(lastpx >> 8, run / 3u);
                col += run;
                if (col == stride) col = 0, row++;
            } else {
                if (kind == kTokEob)
                    err |= kEmitSawEob, eob_end = pos + used;
                else
                    err = kEmitBadStream;
                stop = true;
                break;
            }
            if (used > 14 || pos + used >= limit) break;
        }
        pos += used;
    }
    out.finish();
    return err;
}

// The four literal bytes in front of subsequence g: collected backwards over its predecessors' (literal count, tail) records.
// info(k) / tail(k): records of the file's subsequence k.
template <class Info, class Tail> FPNG_DEC_HD uint32_t lookback_lastpx(uint32_t g, const Info &info, const Tail &tail)
{
    uint32_t v = 0, got = 0; // got bytes collected, the most recent one in bits 31..24
    while (g > 0 && got < 4) {
        g--;
        const uint32_t l = info_lits(info(g));
        if (!l) continue;
        const uint32_t m = l < 4 - got ? l : 4 - got;                 // take its m most recent bytes
        const uint64_t top = (uint64_t)tail(g) >> (32 - 8 * m);       // (m = 4: the whole word)
        v |= (uint32_t)(top << (32 - 8 * (got + m)));
        got += m;
    }
    return v;
}

} // namespace dec
} // namespace fpng_amd
