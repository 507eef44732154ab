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
constexpr uint32_t kMergeSteps = 24; // sub_refix: tokens stepped one by one before it decodes the subsequence again as a whole
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

// the low `bits` bits of v (bits <= 24; 0 -> 0)
FPNG_DEC_HD uint32_t low_bits(uint32_t v, uint32_t bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, 0u, bits);
#else
    return v & ((1u << bits) - 1u);
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

// When does a lane whose next token needs the general path (fetch()) get it?  On the GPU the lanes of a wave run in lockstep: a
// token that is rare for one lane (a long match, a group of literals that reaches over the limit) turns up in SOME lane of the
// wave almost every iteration, and the whole wave walks through the general path each time.  So such a lane waits -- the
// straight-line part of the loop does nothing for it -- until eight lanes wait or no lane can go on (WaveVote, decode.hip).
// On the host every lane is alone.
struct VoteAlone {
    static FPNG_DEC_HD bool go(bool waiting) { return waiting; }
};

// Decodes the tokens that start in [pos, limit) (positions: bits relative to the staged slice); returns the position behind the
// last one.  Written for the SIMT machine: one iteration reads a 32-bit window and does TWO lookups (the second one on the bits
// behind the first token) as straight-line predicated code -- a group of literals that lies wholly in front of the limit and a
// match are applied -- and what is left (the end of the block, an invalid code, a group that reaches over the limit) takes a
// single token through fetch(), when the vote says so.
template <bool Count, class Vote, class Bits>
FPNG_DEC_HD uint32_t walk_count(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t pos, uint32_t limit, uint32_t data_limit, SubCount &c)
{
    uint32_t lits = c.lits, tail = c.tail, runs = 0, flags = 0;
    const uint32_t lim = limit < data_limit ? limit : data_limit; // (no token may start at or behind data_limit)
    // one token at the window's first bit if it is a plain one; returns the bits it took (0: not a plain one)
    auto take = [&](uint32_t wk, uint32_t room, bool en) -> uint32_t {
        const uint32_t e = lut[wk & (kLutEntries - 1)], L = e >> 28, n = (e >> 26) & 3u, xb = (e >> 9) & 7u;
        const bool lit = en && n != 0 && L <= room, mt = en && n == 0 && (e & kEntMatch) != 0;
        if (Count) {
            lits += lit ? n : 0u;
            tail = funnel(e & 0xFFFFFFu, tail, lit ? 8 * n : 0u);
            runs += mt ? (e & 511u) + ((wk >> L) & ((1u << xb) - 1u)) : 0u;
        }
        return lit ? L : (mt ? L + xb + 1 : 0u);
    };
    while (pos < lim) {
        const uint32_t w = in.window(pos), room = lim - pos;
        const uint32_t ba = take(w, room, true);
        const bool en_b = ba != 0 && ba <= 14 && ba < room; // (the second token starts inside, with 18 valid bits in the window)
        const uint32_t bb = take(w >> ba, room - ba, en_b);
        pos += ba + bb;
        const bool waiting = pos < lim && (ba == 0 || (en_b && bb == 0)); // the token at pos is not a plain one
        if (Vote::go(waiting)) {
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
template <class Vote, class Bits>
FPNG_DEC_HD void sub_first(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t lead_start, uint32_t nominal, uint32_t boundary, uint32_t data_limit, SubState &s)
{
    uint32_t p = nominal;
    if (lead_start < nominal) {
        SubCount d = {0, 0, 0, 0};
        p = walk_count<false, Vote>(in, lut, lenof, lead_start, nominal, data_limit, d);
        if (d.flags || p < nominal) p = nominal; // the lead-in derailed: any start is as good as another
    }
    s.start = p;
    s.c.bytes = s.c.lits = s.c.tail = s.c.flags = 0;
    const uint32_t e = walk_count<true, Vote>(in, lut, lenof, p, boundary, data_limit, s.c);
    // A decode that derailed or met an end-of-block symbol hands over on the nominal boundary: speculative decodes meet FALSE
    // end-of-block symbols; which one is the true one is settled afterwards (the first one of the chain).
    s.end = s.c.flags ? boundary : e;
}

// The subsequence must start at `want` instead of s.start (its predecessor ended there).  Both decodes -- the old one from
// s.start, the new one from `want` -- are stepped token by token, the one that lags behind first; where they meet, the rest of the
// old decode holds, and only the counts in front of that point are exchanged.  No meeting point inside the subsequence (or one
// so late that the last four literals are not all behind it): decoded again as a whole.
template <class Vote, class Bits>
FPNG_DEC_HD void sub_refix(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t want, uint32_t boundary, uint32_t data_limit, SubState &s)
{
    uint32_t A = s.start, B = want;
    SubCount a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    // (at most kMergeSteps tokens: two decodes that have not met by then rarely will -- a periodic stream keeps them apart for
    //  good -- and stepping token by token costs several times the group-wise walk that follows)
    for (uint32_t steps = 0; A != B && steps < kMergeSteps; steps++) {
        if ((A < B ? A : B) >= boundary) break;
        if (A < B) {
            A = walk_count<true, VoteAlone>(in, lut, lenof, A, A + 1, data_limit, a);
            if (a.flags) break;
        } else {
            B = walk_count<true, VoteAlone>(in, lut, lenof, B, B + 1, data_limit, b);
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
    const uint32_t e = walk_count<true, Vote>(in, lut, lenof, want, boundary, data_limit, s.c);
    s.end = s.c.flags ? boundary : e;
}

// ... decoded again as a whole (where the phase maps hand a thread its true start: nearly every thread of the workgroup changes then)
template <class Vote, class Bits>
FPNG_DEC_HD void sub_redo(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t want, uint32_t boundary, uint32_t data_limit, SubState &s)
{
    s.start = want;
    s.c.bytes = s.c.lits = s.c.tail = s.c.flags = 0;
    const uint32_t e = walk_count<true, Vote>(in, lut, lenof, want, boundary, data_limit, s.c);
    s.end = s.c.flags ? boundary : e;
}

// ---- phase maps: the way out of a PERIODIC stream ----
// A stream that repeats itself (flat or striped content: every row the same few tokens) keeps wrongly started decoders in a stable
// false phase -- they never fall into step -- and the corrections above then crawl through a workgroup one subsequence per step.
// But a decoder crosses a subsequence's nominal first bit in one of only 18 PHASES (the bit its next token starts at, counted
// from that nominal bit: a token has at most 18 bits), a periodic stream keeps a handful of them alive, and what a subsequence
// does to EACH phase can be found in parallel.  Every thread keeps a map phase -> phase (where a decode of its subsequence that
// starts in phase x ends, as a phase of the NEXT subsequence; kPhaseUnknown where nobody has looked): its own decode's pair, plus
// the phases in which decoders started at 17 more consecutive bits of its lead-in arrive (the true token sequence has a boundary
// among any 18 consecutive bits, so the true phase is among the arrivals), plus -- until no map grows -- the phases its
// predecessor's map ends in.  The maps are then composed along the workgroup (an inclusive prefix "sum" over functions), which
// yields every thread's true start at once and the workgroup's own map (entry -> exit) for the same game one level up
// (dec_chain_kernel).  A map is three dwords: phase x in bits 5 (x % 6) .. of word x / 6.  A decode that derails or meets an
// end-of-block symbol "ends" on its nominal boundary (phase 0), as in SubState.
constexpr uint32_t kPhases = 18, kPhaseUnknown = 31;
#ifndef FPNG_DEC_REFIX_ROUNDS
#define FPNG_DEC_REFIX_ROUNDS 3
#endif
constexpr uint32_t kRefixRounds = FPNG_DEC_REFIX_ROUNDS; // correction steps inside a workgroup before the phase maps take over
constexpr uint32_t kCandGrowSteps = 12; // steps in which the maps take over their neighbours' ends (a bound, not a need: the seeds usually hold every phase)
struct PhaseMap {
    uint32_t w[3];
};
FPNG_DEC_HD PhaseMap pm_none()
{
    PhaseMap m;
    m.w[0] = m.w[1] = m.w[2] = 0x3FFFFFFFu;
    return m;
}
FPNG_DEC_HD uint32_t pm_at(const PhaseMap &m, uint32_t x) // x < kPhases
{
    const uint32_t q = x >= 12u ? 2u : (x >= 6u ? 1u : 0u), r = x - 6u * q;
    return ((q == 0 ? m.w[0] : (q == 1 ? m.w[1] : m.w[2])) >> (5u * r)) & 31u;
}
FPNG_DEC_HD void pm_set(PhaseMap &m, uint32_t x, uint32_t v)
{
    const uint32_t q = x >= 12u ? 2u : (x >= 6u ? 1u : 0u), r = x - 6u * q, keep = ~(31u << (5u * r)), b = v << (5u * r);
    m.w[0] = q == 0 ? (m.w[0] & keep) | b : m.w[0], m.w[1] = q == 1 ? (m.w[1] & keep) | b : m.w[1], m.w[2] = q == 2 ? (m.w[2] & keep) | b : m.w[2];
}
FPNG_DEC_HD PhaseMap pm_one(uint32_t start, uint32_t end)
{
    PhaseMap m = pm_none();
    pm_set(m, start, end);
    return m;
}
FPNG_DEC_HD uint32_t pm_count(const PhaseMap &m) // phases it knows
{
    uint32_t n = 0;
    for (uint32_t x = 0; x < kPhases; x++) n += pm_at(m, x) != kPhaseUnknown;
    return n;
}
// first the span `a` stands for, then `b`
FPNG_DEC_HD PhaseMap pm_compose(const PhaseMap &a, const PhaseMap &b)
{
    PhaseMap r = pm_none();
    for (uint32_t x = 0; x < kPhases; x++) {
        const uint32_t e = pm_at(a, x);
        if (e != kPhaseUnknown) pm_set(r, x, pm_at(b, e));
    }
    return r;
}
// where a decode of [start, boundary) ends, without counting anything
template <class Vote, class Bits>
FPNG_DEC_HD uint32_t sub_probe(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t start, uint32_t boundary, uint32_t data_limit)
{
    SubCount d = {0, 0, 0, 0};
    const uint32_t e = walk_count<false, Vote>(in, lut, lenof, start, boundary, data_limit, d);
    return d.flags ? boundary : e;
}
// One growing step of a thread's map: every phase the predecessor's map ends in (= a phase this subsequence is entered in: its
// boundary is this thread's nominal bit) that the map does not know yet is decoded.  Returns whether the map grew.
template <class Vote, class Bits>
FPNG_DEC_HD bool pm_grow(const Bits &in, const uint32_t *lut, const uint8_t *lenof, const PhaseMap &pred, uint32_t nominal, uint32_t boundary, uint32_t data_limit, PhaseMap &map)
{
    bool grew = false;
    for (uint32_t x = 0; x < kPhases; x++) {
        const uint32_t s = pm_at(pred, x);
        if (s == kPhaseUnknown || pm_at(map, s) != kPhaseUnknown) continue;
        pm_set(map, s, sub_probe<Vote>(in, lut, lenof, nominal + s, boundary, data_limit) - boundary);
        grew = true;
    }
    return grew;
}
// A thread's first map: its own decode's pair is there; added are the phases in which decoders started at the bits lead_start + 1
// ... lead_start + 17 cross the nominal bit.
template <class Vote, class Bits>
FPNG_DEC_HD void pm_seed(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t lead_start, uint32_t nominal, uint32_t boundary, uint32_t data_limit, PhaseMap &map)
{
    for (uint32_t j = 1; j < kPhases; j++) {
        SubCount d = {0, 0, 0, 0};
        const uint32_t p = walk_count<false, Vote>(in, lut, lenof, lead_start + j, nominal, data_limit, d);
        if (d.flags || p < nominal || pm_at(map, p - nominal) != kPhaseUnknown) continue;
        pm_set(map, p - nominal, sub_probe<Vote>(in, lut, lenof, p, boundary, data_limit) - boundary);
    }
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
// 16-BYTE GROUPS only (Sink::store128(group index, four dwords)): the group in which its output ends is completed with the first
// bytes of the following subsequences -- it simply decodes on until the group is full -- and a thread whose output starts inside a
// group leaves that group to the thread in front of it.  No byte stores, nothing is written twice, no store that depends on whose
// bytes a dword holds.  (Sink::store32 serves the one place where the stream itself ends inside a group.)  4-byte stores from 64
// lanes whose ranges lie ~150 bytes apart were 64 memory transactions of 4 bytes per instruction, and 6 x the bytes at the fabric.
enum : uint32_t { kEmitBadStream = 2u, kEmitLeaveToCpu = 16u, kEmitSawEob = 0x100u }; // (= kDecBadStream, kDecStalled, kDecSawEob of decode.h)
constexpr uint32_t kFillMinDwords = 16, kFillMinPixels3 = 32; // runs from this length on are filled by the wave (StreamWriter::run4 / run3)

template <class Sink> struct StreamWriter {
    Sink &sink;
    uint32_t acc;  // bytes not stored yet, the oldest one lowest
    uint32_t have; // how many (0..3)
    uint32_t dw;   // stream dword they go to
    uint32_t q0, q1, q2, q3; // the last whole dwords, the newest in q3: at the end of a 16-byte group they are its four dwords
    bool skip;     // the group the output starts in belongs to the thread in front (until its end is reached)
    FPNG_DEC_HD StreamWriter(Sink &s, uint64_t off) : sink(s), acc(0), have((uint32_t)off & 3u), dw((uint32_t)(off >> 2)), q0(0), q1(0), q2(0), q3(0)
    {
        skip = ((uint32_t)off & 15u) != 0;
    }
    FPNG_DEC_HD void push(uint32_t v, bool full) // a whole dword (where `full`)
    {
        q0 = full ? q1 : q0, q1 = full ? q2 : q1, q2 = full ? q3 : q2, q3 = full ? v : q3;
        const bool gend = full && (dw & 3u) == 3u;
        if (gend && !skip) sink.store128(dw >> 2, q0, q1, q2, q3);
        skip = skip && !gend;
        dw += full;
    }
    FPNG_DEC_HD void put(uint32_t bytes, uint32_t n) // n <= 4 bytes, the first one lowest; bytes = 0 where n = 0
    {
        const uint32_t sh = 8 * have, lo = acc | (bytes << sh), hi = (bytes >> 8) >> (24 - sh), nh = have + n;
        const bool full = nh >= 4;
        push(lo, full);
        acc = full ? hi : lo;
        have = nh & 3u;
    }
    // npix copies of a 4-byte pixel: the first dword completes the pending bytes, the others are one rotated constant.
    // A LONG run is not this thread's to write dword by dword -- its wave would wait for it, and on flat content (a screenshot:
    // a few dozen bytes of tokens stand for kilobytes of pixels) every thread has such runs: the thread writes up to the next
    // 16-byte group boundary and hands the run's whole groups to Sink::fill(first group, groups, three dwords d0 d1 d2: the
    // dwords from there on are d0 d1 d2 d0 ...), which the wave's threads store together, 64 groups per instruction
    // (Sink::cooperate(), called by walk_emit once per iteration).
    FPNG_DEC_HD void run4(uint32_t px, uint32_t npix)
    {
        const uint32_t sh = 8 * have, lo = px << sh, hi = (px >> 8) >> (24 - sh), r = lo | hi;
        push(acc | lo, true);
        uint32_t left = npix - 1;
        if (left >= kFillMinDwords) {
            while (dw & 3u) push(r, true), left--; // (its last push closed a group: `skip` is off from here)
            const uint32_t groups = left >> 2;
            sink.fill(dw >> 2, groups, r, r, r);
            dw += 4 * groups, left -= 4 * groups;
            q0 = q1 = q2 = q3 = r;
        }
        for (uint32_t j = 0; j < left; j++) push(r, true);
        acc = hi;
    }
    // ... of a 3-byte pixel: the stream repeats every 3 bytes, its dwords every 3 dwords; three groups = 48 bytes = 16 pixels
    // leave the writer where it was (pending bytes, place inside the pixel)
    FPNG_DEC_HD void run3(uint32_t px, uint32_t npix)
    {
        uint32_t left = npix;
        if (npix >= kFillMinPixels3) {
            for (uint32_t k = 0; k < 4; k++) put(px, 3); // (12 bytes: the pending bytes are the run's own from here on)
            left -= 4;
            while (dw & 3u) put(px, 3), left--; // (at most 5 pixels; the last push closed a group)
            uint32_t groups = (3 * left) >> 4;
            groups -= groups % 3u;
            // byte 0 of dword dw is the pixel's byte (3 - have) % 3: acc holds the last `have` bytes of a pixel
            const uint64_t wrap = (uint64_t)px | (uint64_t)px << 24 | (uint64_t)px << 48;
            const uint32_t ph = (3u - have) % 3u;
            const uint32_t d0 = (uint32_t)(wrap >> (8 * ph)), d1 = (uint32_t)(wrap >> (8 * ((ph + 1) % 3u))), d2 = (uint32_t)(wrap >> (8 * ((ph + 2) % 3u)));
            sink.fill(dw >> 2, groups, d0, d1, d2);
            if (groups) q0 = d2, q1 = d0, q2 = d1, q3 = d2; // (the last group of a multiple of three)
            dw += 4 * groups, left -= 16 * (groups / 3u);
        }
        for (uint32_t k = 0; k < left; k++) put(px, 3);
    }
    // Only where the stream ends inside a group: its whole dwords, and the pending bytes (zeros behind them: the buffer is padded).
    FPNG_DEC_HD void finish()
    {
        if (have) push(acc, true), have = 0;
        const uint32_t m = dw & 3u; // dwords 0..m-1 of the group are in q[4-m]..q3
        if (skip) return;
        if (m >= 3) sink.store32(dw - 3, q1);
        if (m >= 2) sink.store32(dw - 2, q2);
        if (m >= 1) sink.store32(dw - 1, q3);
    }
};

// The real decode of one subsequence.  The synchronisation has settled where it starts (pos) and how many bytes its tokens stand
// for (own): the loop is driven by the BYTE count -- `own` bytes plus the `pad` (0..15) first bytes of the following subsequences
// that complete its last 16-byte group -- so no token has to be compared with a bit limit and a group of literals is simply cut to the
// bytes still wanted.  off = stream byte of its first output byte, col = that byte's place in its row (0 = the filter byte),
// lastpx = the four literal bytes in front of it; C = channels in the file; last: the stream's last subsequence (an end-of-block
// symbol must follow its bytes).  Two predicated lookups per window: literals and matches of exactly ONE pixel (the usual kind on
// noisy content: length C, no extra bits) are applied in line, everything else takes the branch.  Checked here, token by token:
// a match repeats whole pixels, starts on a pixel and stays inside its row (reference src/fpng.cpp:2273-2330); that every row
// starts with its filter literal is checked where the rows are read (dec_unfilter_kernel).  Returns kEmit* flags; eob_end =
// position behind the end-of-block symbol.
template <int C, class Bits, class Sink>
FPNG_DEC_HD uint32_t walk_emit(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t pos, uint32_t own, uint32_t pad, bool last, uint64_t off, uint32_t col,
                               uint32_t lastpx, uint32_t stride, Sink &sink, uint32_t &eob_end)
{
    StreamWriter<Sink> out(sink, off);
    const uint32_t bpl = stride - 1;
    uint32_t todo = own + pad;
    uint32_t rowleft = stride - col; // bytes up to the end of the row, the next one included (== stride: the next byte is a filter byte)
    uint32_t err = 0;
    // one token at the window's first bit if it is a plain one; returns the bits it took (0: not a plain one, or nothing left to do)
    // (the straight-line part keeps rl = rowleft - 1, 0 .. stride - 1: "wrap at the row's end" is then min(t, t + stride) of
    //  t = rl - bytes taken -- the unsigned difference is huge exactly when the row ended; the match branch below works on rowleft)
    uint32_t rl = rowleft - 1;
    auto take = [&](uint32_t wk) -> uint32_t {
        const uint32_t e = lut[wk & (kLutEntries - 1)], L = e >> 28, n = (e >> 26) & 3u;
        const bool m1 = (e & 0x0FFFFFFFu) == (kEntMatch | (uint32_t)C) && (rl + 1) % C == 0 && rl != bpl - 1 && todo >= (uint32_t)C;
        const uint32_t nl = n < todo ? n : todo, nb = m1 ? (uint32_t)C : nl;
        const uint32_t lits = low_bits(e, 8 * nl); // (nl = 0: no byte)
        out.put(m1 ? (C == 4 ? lastpx : lastpx >> 8) : lits, nb);
        lastpx = funnel(lits, lastpx, m1 ? 0u : 8 * nl);
        const uint32_t t = rl - nb, t2 = t + stride;
        rl = t < t2 ? t : t2;
        todo -= nb;
        return nb ? L + (m1 ? 1u : 0u) : 0u;
    };
    while (sink.any(todo != 0)) { // (the whole wave stays until its last thread is done: Sink::cooperate() needs them all)
        const uint32_t w = in.window(pos);
        const uint32_t ba = take(w);
        const uint32_t bb = take(w >> ba); // (a first token that was not plain is looked at again, to no effect)
        pos += ba + bb;
        if (todo && !bb) { // the token at pos is not a plain one: a match (or the stream ends, or derails, with bytes still owed)
            rowleft = rl + 1;
            // Matches that follow one another repeat the same pixel (no literal in between, and none of them may leave its row):
            // they are written as ONE run -- a flat row of a screenshot is a few dozen maximal matches.
            const uint32_t px = C == 4 ? lastpx : lastpx >> 8;
            uint32_t whole = 0; // bytes of the matches that lie wholly inside this thread's bytes
            bool stop = false;
            for (;;) {
                uint32_t n3, l3 = 0, run = 0, bits;
                const uint32_t kind = fetch(in.window(pos), lut, lenof, 64u, n3, l3, run, bits);
                const bool mine = todo > pad; // (the tokens behind this thread's bytes are their owners' to check)
                if (kind != kTokMatch) {
                    // behind a match: an ordinary token, the loop's next iteration takes it; else the stream ends, or derails, with bytes still owed
                    if (!whole) stop = true, err |= mine ? kEmitBadStream : 0u;
                    break;
                }
                if (mine && (rowleft % C != 0 || rowleft > bpl || run > rowleft || run % C != 0)) {
                    stop = true, err |= kEmitBadStream;
                    break;
                }
                // A match at a row's FIRST pixel repeats a pixel of zeros (reference :2268: prev_delta_* start at 0), and so do the
                // matches behind it until a literal pixel comes -- but "the last literal bytes" this decoder keeps know nothing
                // of rows.  fpng's encoders never write such a match (a row's first pixel has no left neighbour); a file that
                // has one is left to the CPU decoder.
                if (mine && rowleft == bpl) {
                    stop = true, err |= kEmitLeaveToCpu;
                    break;
                }
                pos += bits;
                const uint32_t r = run < todo ? run : todo;
                rowleft -= r;
                rowleft += (int32_t)rowleft <= 0 ? stride : 0u;
                todo -= r;
                if (r == run && mine)
                    whole += run;
                else { // the match reaches into (or lies in) the pad: byte by byte
                    if (whole) C == 4 ? out.run4(px, whole >> 2) : out.run3(px, whole / 3u);
                    whole = 0;
                    for (uint32_t k = 0; k < r; k++) out.put((px >> (8 * (k % C))) & 255u, 1);
                    break;
                }
                if (!todo || rowleft == stride) break; // (a row ended: a filter literal must follow)
            }
            if (whole) C == 4 ? out.run4(px, whole >> 2) : out.run3(px, whole / 3u);
            rl = rowleft - 1;
            if (stop) break;
        }
        sink.cooperate();
    }
    if (last && !err) { // the end-of-block symbol
        uint32_t n3, l3 = 0, run = 0, bits;
        const uint32_t kind = fetch(in.window(pos), lut, lenof, 64u, n3, l3, run, bits);
        if (kind == kTokEob)
            err |= kEmitSawEob, eob_end = pos + bits;
        else
            err |= kEmitBadStream;
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
