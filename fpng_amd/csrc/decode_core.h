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
// last one.  Written for the SIMT machine: one iteration reads a 32-bit window and does TWO lookups (the second one on the bits
// behind the first one's group) as straight-line predicated code -- a group of literals that lies wholly in front of the limit is
// applied, anything else (a match, the end of the block, an invalid code, a group that reaches over the limit) is left to ONE
// branch at the iteration's end, which takes a single token through fetch().
template <bool Count, class Bits>
FPNG_DEC_HD uint32_t walk_count(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t pos, uint32_t limit, uint32_t data_limit, SubCount &c)
{
    uint32_t lits = c.lits, tail = c.tail, runs = 0, flags = 0;
    const uint32_t lim = limit < data_limit ? limit : data_limit; // (no token may start at or behind data_limit)
    while (pos < lim) {
        const uint32_t w = in.window(pos);
        const uint32_t ea = lut[w & (kLutEntries - 1)], la = ea >> 28, na = (ea >> 26) & 3u, room_a = lim - pos;
        const bool ca = na != 0 && la <= room_a;
        const uint32_t eb = lut[(w >> la) & (kLutEntries - 1)], lb = eb >> 28, nb = (eb >> 26) & 3u, room_b = room_a - la;
        const bool cb = ca && nb != 0 && lb <= room_b;
        const uint32_t n1 = ca ? na : 0u, n2 = cb ? nb : 0u;
        if (Count) {
            lits += n1 + n2;
            tail = funnel(ea & 0xFFFFFFu, tail, 8 * n1);
            tail = funnel(eb & 0xFFFFFFu, tail, 8 * n2);
        }
        pos += (ca ? la : 0u) + (cb ? lb : 0u);
        if (!cb && pos < lim) { // the token at pos is not a plain group of literals
            uint32_t n3, l3 = 0, run = 0, bits;
            const uint32_t kind = fetch(in.window(pos), lut, lenof, lim - pos, n3, l3, run, bits);
            if (kind >= kTokEob) {
                flags = kind == kTokEob ? kSubEob : kSubInvalid;
                break;
            }
            pos += bits;
            if (Count) {
                if (kind == kTokLit)
                    lits += n3, tail = funnel(l3, tail, 8 * n3);
                else
                    runs += run;
            }
        }
    }
    if (!flags && pos < limit) flags = kSubInvalid; // ran off the data without an end-of-block symbol
    c.flags = flags;
    if (Count) c.bytes += (lits - c.lits) + runs, c.lits = lits, c.tail = tail;
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

// ---- the real decode: one subsequence's tokens into the filtered stream ----
// The filtered stream = what the reference's decoder consumes row by row (src/fpng.cpp:2255-2262): h rows of 1 filter byte +
// w * c bytes, kept in exactly this layout (the column kernels read it with unaligned loads).  Every thread stores WHOLE ALIGNED
// DWORDS only: the dword in which its output ends is completed with the first bytes of the following subsequences -- it simply
// decodes on until the dword is full -- and a thread whose output starts inside a dword leaves that dword to the thread in front
// of it.  No byte stores, no dword is written twice.  Four dwords at a time where a 16-byte group of the stream is all the
// thread's (Sink::store128(group index, four values)), single dwords at its two ends (Sink::store32(dword index, value)):
// 4-byte stores from 64 lanes whose ranges lie ~150 bytes apart are 64 memory transactions of 4 bytes each.
struct EmitGeom {
    uint32_t stride; // w * c + 1
    uint32_t c;      // channels in the file
};
enum : uint32_t { kEmitBadStream = 2u, kEmitSawEob = 0x100u };

template <class Sink> struct StreamWriter {
    Sink &sink;
    uint32_t acc;  // bytes not stored yet, the oldest one lowest
    uint32_t have; // how many (0..3)
    uint32_t dw;   // stream dword they go to
    uint32_t q0, q1, q2, q3; // the last whole dwords, the newest in q3: at the end of a 16-byte group they are its four dwords
    uint32_t own;  // first dword of the current group that is this thread's to store (0 except in its first group)
    FPNG_DEC_HD StreamWriter(Sink &s, uint64_t off) : sink(s), acc(0), have((uint32_t)off & 3u), dw((uint32_t)(off >> 2)), q0(0), q1(0), q2(0), q3(0)
    {
        own = (dw & 3u) + (have != 0); // (a first dword that starts in front of `off` belongs to the thread in front)
    }
    FPNG_DEC_HD void push(uint32_t v, bool full) // a whole dword (where `full`)
    {
        q0 = full ? q1 : q0, q1 = full ? q2 : q1, q2 = full ? q3 : q2, q3 = full ? v : q3;
        if (full && (dw & 3u) == 3u) {
            if (!own)
                sink.store128(dw >> 2, q0, q1, q2, q3);
            else {
                if (own <= 1) sink.store32(dw - 2, q1);
                if (own <= 2) sink.store32(dw - 1, q2);
                if (own <= 3) sink.store32(dw, q3);
                own = 0;
            }
        }
        dw += full;
    }
    FPNG_DEC_HD void put(uint32_t bytes, uint32_t n) // n <= 3 bytes, the first one lowest; bytes = 0 where n = 0
    {
        const uint32_t sh = 8 * have, lo = acc | (bytes << sh), hi = (bytes >> 8) >> (24 - sh), nh = have + n;
        const bool full = nh >= 4;
        push(lo, full);
        acc = full ? hi : lo;
        have = nh & 3u;
    }
    // npix copies of a 4-byte pixel: the first dword completes the pending bytes, the others are one rotated constant
    FPNG_DEC_HD void run4(uint32_t px, uint32_t npix)
    {
        const uint32_t sh = 8 * have, lo = px << sh, hi = (px >> 8) >> (24 - sh), r = lo | hi;
        push(acc | lo, true);
        for (uint32_t j = 1; j < npix; j++) push(r, true);
        acc = hi;
    }
    FPNG_DEC_HD void run3(uint32_t px, uint32_t npix)
    {
        for (uint32_t k = 0; k < npix; k++) put(px, 3);
    }
    FPNG_DEC_HD void finish() // the thread's end: the whole dwords of its last, unfinished group; at the stream's end also the pending bytes (zeros behind them: the buffer is padded)
    {
        if (have) push(acc, true), have = 0;
        const uint32_t m = dw & 3u; // dwords 0..m-1 of the group are in q[4-m]..q3
        if (m >= 3 && own <= 0) sink.store32(dw - 3, q1);
        if (m >= 2 && own <= m - 2) sink.store32(dw - 2, q2);
        if (m >= 1 && own <= m - 1) sink.store32(dw - 1, q3);
    }
};

// pos / limit / data_limit as in walk_count; off = stream byte of the subsequence's first output byte, (row, col) = its place in
// the image (col 0 = the filter byte), lastpx = the four literal bytes in front of it.  Returns kEmit* flags; eob_end = position
// behind the end-of-block symbol if it met one.  Same shape as walk_count: two predicated lookups per window -- a group of
// literals in front of the limit and strictly inside its row is applied -- and one branch for everything else.
template <class Bits, class Sink>
FPNG_DEC_HD uint32_t walk_emit(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t pos, uint32_t limit, uint32_t data_limit, uint64_t off, uint32_t row,
                               uint32_t col, uint32_t lastpx, const EmitGeom &g, Sink &sink, uint32_t &eob_end)
{
    StreamWriter<Sink> out(sink, off);
    if (col == 1) lastpx = 0; // right behind a filter byte there is no previous pixel: zeros (reference :2262 prev_delta_* = 0)
    uint32_t err = 0;
    const uint32_t stride = g.stride, c = g.c;
    const uint32_t lim = limit < data_limit ? limit : data_limit;
    bool over = false; // behind the limit: only the last dword is being completed, the tokens are their owners' to check
    for (;;) {
        if (pos >= lim) {
            if (!over && pos < limit) err |= kEmitBadStream; // ran off the data
            over = true;
            if (!out.have || pos >= data_limit) break;
        }
        uint32_t n3 = 0, l3 = 0, run = 0, bits = 0, kind = kTokLit;
        if (!over) {
            const uint32_t w = in.window(pos);
            const uint32_t ea = lut[w & (kLutEntries - 1)], la = ea >> 28, na = (ea >> 26) & 3u, room_a = lim - pos;
            const bool ca = na != 0 && la <= room_a && col != 0 && col + na < stride;
            const uint32_t n1 = ca ? na : 0u, lits1 = ca ? (ea & 0xFFFFFFu) : 0u;
            out.put(lits1, n1);
            lastpx = funnel(lits1, lastpx, 8 * n1);
            col += n1;
            const uint32_t eb = lut[(w >> la) & (kLutEntries - 1)], lb = eb >> 28, nb = (eb >> 26) & 3u, room_b = room_a - la;
            const bool cb = ca && nb != 0 && lb <= room_b && col + nb < stride;
            const uint32_t n2 = cb ? nb : 0u, lits2 = cb ? (eb & 0xFFFFFFu) : 0u;
            out.put(lits2, n2);
            lastpx = funnel(lits2, lastpx, 8 * n2);
            col += n2;
            pos += (ca ? la : 0u) + (cb ? lb : 0u);
            if (cb || pos >= lim) continue;
            kind = fetch(in.window(pos), lut, lenof, lim - pos, n3, l3, run, bits);
        } else {
            kind = fetch(in.window(pos), lut, lenof, 64u, n3, l3, run, bits);
            const uint32_t need = 4 - out.have; // bytes that complete the dword
            if (kind == kTokLit && n3 > need) n3 = need;
            if (kind == kTokMatch && run > need) { // the first bytes of the repeated pixel (a match starts on a pixel)
                kind = kTokLit, n3 = need;
                l3 = (c == 4 ? lastpx : lastpx >> 8) & (0xFFFFFFu >> (8 * (3 - need)));
            }
        }
        pos += bits;
        if (kind == kTokLit) {
            for (uint32_t j = 0; j < n3; j++) {
                const uint32_t b = (l3 >> (8 * j)) & 255u;
                out.put(b, 1);
                if (!col) { // the row's filter literal: 0, then 2 = Up (reference :2255-2259)
                    if (!over && b != (row ? 2u : 0u)) err |= kEmitBadStream;
                    lastpx = 0, col = 1;
                } else {
                    lastpx = funnel(b, lastpx, 8);
                    if (++col == stride) col = 0, row++;
                }
            }
        } else if (kind == kTokMatch) {
            // a match repeats the previous pixel: whole pixels, starting on a pixel, inside the row (reference :2273-2330)
            const uint32_t x = col - 1;
            const bool whole = c == 4 ? !((x | run) & 3u) : (x % 3u == 0 && run % 3u == 0);
            if (!col || !whole || col + run > stride) {
                if (!over) err |= kEmitBadStream;
                break;
            }
            if (c == 4)
                out.run4(lastpx, run >> 2);
            else
                out.run3(lastpx >> 8, run / 3u);
            col += run;
            if (col == stride) col = 0, row++;
        } else {
            if (!over) {
                if (kind == kTokEob)
                    err |= kEmitSawEob, eob_end = pos;
                else
                    err |= kEmitBadStream;
            }
            break;
        }
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
