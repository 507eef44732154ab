// decode_core.h -- the per-thread logic of the GPU decoder (decode.hip), written so that it also compiles for the host:
// tests/cpp/decode_emul.cpp runs the same walkers thread by thread on the CPU (no GPU in the dev container), and
// tools/sync_stats.c measured the numbers the constants come from.  Reference: src/fpng.cpp:2209-2901 decodes the same streams
// serially, one token at a time; here a stream is cut into subsequences of kSubBits token bits that are decoded independently.
//
// THE LOOKUP TABLE (built on the host, decode_api.cpp: build_multi_lut, or by dec_build_lut_kernel).  Index = the next 12 stream
// bits, one 32-bit entry.  A SIMPLE token -- a group of literals, or a match WITHOUT extra bits (lengths 3 .. 10 and 258: one, two,
// three pixels, the matches of noisy content; whole runs) -- has:
//   bits 31..28  the stream bits it takes, ALL of them (1..13: a match's 1-bit distance code -- "the previous pixel", reference
//                src/fpng.cpp:2301 -- included); 0: not a simple token
//   bits 27..26  n: number of LITERALS decoded at once (1..3: as many whole literal codes as fit into the 12 bits, at most 3);
//                their byte values in bits 7..0, 15..8, 23..16 (first one lowest)
//   n == 0:      bit 25 (kEntMatch) set, bits 8..0 the match's length
// -- the entry without its upper four bits is the token's RECORD (below).  Every other token:
//   a match with extra bits: kEntMatch, bits 8..0 base length (11..227), bits 11..9 number of extra bits (1..5), bits 15..12 the
//                length symbol's code bits (the extra bits and the 1-bit distance code follow in the stream)
//   the end of the block:    kEntEob, bits 15..12 its code bits
//   no such code:            0
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
constexpr uint32_t kEntEob = 1u << 24;
constexpr uint32_t kEntSimpleMin = 1u << 28; // (an entry at or above it is a simple token's)
// TOKEN RECORDS (round 6: every token is decoded ONCE).  The decode that settles a subsequence leaves what it decoded behind, and the
// pass that writes the pixels (dec_unfilter_kernel) reads records instead of Huffman codes.  A record is 32 bits:
//   a group of literals   n << 26 | the n bytes (first one lowest)          -- the table's entry without its length field
//   a match               kRecRun | its length in bytes (3..258) in bits 23..0
//   0                     nothing
// and an ENTRY is two of them -- what one step of the walk (two lookups) decoded, stored with one 8-byte store; a token that takes the
// walk's general path is an entry of its own.  A subsequence has room for kRecCap entries (its 512 bits in steps of four bits and
// more); one that needs more is flagged and its file left to the CPU decoder (no fpng encoder's output comes near: a step of two
// groups of three literals takes six bits at the very least).
// LAYOUT in memory, made for both sides.  The writers are the 64 lanes of a wave, one subsequence each, walking in step: entry k of
// all of them at once.  The readers (dec_unfilter_kernel's tiles) are EIGHT neighbouring subsequences at a time -- eight lanes per
// row of a tile, the tile's rows far apart in the stream -- reading their entries k, k + 1, ... in batches.  So a chunk (64
// subsequences) is cut into blocks of 4 entries x 8 subsequences = 256 bytes: entry k of subsequence l (of the chunk) is 8-byte word
//     (k / 4) * 256 + (l / 8) * 32 + (k % 4) * 8 + l % 8
// -- a wave's store of one entry fills eight 64-byte pieces, four steps fill the eight 256-byte blocks whole; a reader's eight lanes
// find four consecutive entries in one block: whole 128-byte lines in both directions.  (Round 6's first layout, entry k of lane l at
// k * 64 + l, had the readers use 64 bytes of every line they fetched, the other half going to another tile on another XCD: 7 GB
// fetched for 1.6 GB of entries, profiles/r06_decode_once_ab.txt.)
constexpr uint32_t kRecCap = 128;
constexpr uint32_t kRecRows = kRecCap + 4;  // + the entry that takes what overflows (a whole block of four)
constexpr uint32_t kRecRun = 1u << 25;      // (= kEntMatch: a table entry of a group of literals never has the bit)
constexpr uint32_t kRecLane = 64;           // subsequences in a chunk
constexpr uint32_t kRecChunk = kRecRows * kRecLane; // 8-byte words of a chunk
FPNG_DEC_HD uint64_t rec_index(uint32_t g, uint32_t k) // 8-byte-word index of entry k of subsequence g
{
    const uint32_t l = g % kRecLane;
    return (uint64_t)(g / kRecLane) * kRecChunk + (k >> 2) * 256u + (l >> 3) * 32u + (k & 3u) * 8u + (l & 7u);
}
enum : uint32_t { kSubEob = 1u, kSubOverflow = 2u, kSubInvalid = 4u }; // (kSubOverflow: more than kRecCap records)
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
    const uint32_t e = lut[w & (kLutEntries - 1)], adv = e >> 28;
    n = (e >> 26) & 3u;
    bits = adv;
    if (n) {
        lits = e & 0xFFFFFFu;
        if (adv > room) { // the group reaches over the stopping bit: token by token
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
        const uint32_t L = adv ? adv - 1u : (e >> 12) & 15u, xb = (e >> 9) & 7u;
        run = (e & 511u) + ((w >> L) & ((1u << xb) - 1u));
        bits = L + xb + 1; // extra bits + the 1-bit distance code ("the previous pixel": reference src/fpng.cpp:2301)
        return kTokMatch;
    }
    bits = (e >> 12) & 15u;
    return (e & kEntEob) ? kTokEob : kTokInvalid;
}

// what a subsequence's decode leaves behind
struct SubCount {
    uint32_t bytes; // output bytes of its tokens
    uint32_t flags; // kSubEob: it met an end-of-block symbol; kSubInvalid: its decode derailed
    uint32_t eob;   // kSubEob: the position behind that symbol
    uint32_t gen;   // matches that took the general path (a walk without Rich: matches with extra bits)
};
// where a walk leaves its records: put2(a, b) -- one step's two records, an entry unless both are nothing.  NoRec: a walk that only
// looks (the lead-in, the phase maps' probes).
struct NoRec {
    FPNG_DEC_HD void put2(uint32_t, uint32_t) {}
    FPNG_DEC_HD uint32_t count() const { return 0; }
};

// When does a lane whose next token needs the general path (fetch()) get it?  At once (VoteAlone): the lanes of a wave that do not
// need it wait for those that do, as the SIMT machine has it.  (Rounds 4-5 also had a wave-wide vote -- such a lane waited until
// eight lanes waited or no lane could go on -- which lost to this on every content once the walk's straight-line part took the
// frequent tokens; the parameter stays: the walks are written against it.)
struct VoteAlone {
    static FPNG_DEC_HD bool go(bool waiting) { return waiting; }
};

FPNG_DEC_HD uint32_t ent_out_bytes(uint32_t e) { return ((e >> 26) & 3u) + ((e & kEntMatch) ? (e & 511u) : 0u); } // a simple token's output bytes (0: none)
constexpr uint32_t kFastRoom = 26; // (a simple token takes 13 bits at most)

// Decodes the tokens that start in [pos, limit) (positions: bits relative to the staged slice); returns the position behind the
// last one.  Written for the SIMT machine: one iteration reads a 32-bit window and does TWO lookups (the second one on the bits
// behind the first token) as straight-line predicated code -- a group of literals that lies wholly in front of the limit and a
// match are applied -- and what is left (the end of the block, an invalid code, a group that reaches over the limit) takes a
// single token through fetch(), when the vote says so.
template <bool Count, bool Rich, class Vote, class Bits, class Rec>
FPNG_DEC_HD uint32_t walk_count(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t pos, uint32_t limit, uint32_t data_limit, SubCount &c, Rec &rec)
{
    uint32_t bytes = 0, flags = 0;
    const uint32_t lim = limit < data_limit ? limit : data_limit; // (no token may start at or behind data_limit)
    // one token at the window's first bit if it is one the straight-line part takes: a group of literals that lies wholly in front of
    // the limit, or a match; returns the bits it took (0: not one of those); r: its record.  A match WITH extra bits (rare in
    // gradients and photographs, frequent where flat runs end at block edges: UI content, tiles) has its length completed under a
    // branch of its own in the RICH form of the walk; in the lean form it goes through fetch(), a token to itself and an iteration
    // of the loop.  Which form: the caller's choice -- the lean one is 4 .. 9 % faster where such matches are rare (the branch sits in
    // the chain lookup -> bits -> next lookup), the rich one 1.3 .. 3 x faster where they are frequent (profiles/r08_sync_matches_ab.txt);
    // dec_sync_kernel looks at what the lead-ins of a wave met.  (Round 5's form cut the extra bits out of the window for every lane
    // and both lookups: 125 vector instructions a step.)
    auto take = [&](uint32_t wk, uint32_t room, bool en, uint32_t &r) -> uint32_t {
        const uint32_t e = lut[wk & (kLutEntries - 1)], adv = e >> 28, n = (e >> 26) & 3u;
        bool ok = en && adv != 0 && (n == 0 || adv <= room);
        uint32_t len = e & 511u, bits = adv;
        if (Rich && en && !adv && (e & kEntMatch)) {
            const uint32_t L = (e >> 12) & 15u, xb = (e >> 9) & 7u; // (18 bits at most with the code and the distance bit: the window has them, see en_b)
            len += (wk >> L) & ((1u << xb) - 1u);
            bits = L + xb + 1u;
            ok = true;
        }
        if (Count) {
            bytes += ok ? (n ? n : len) : 0u;
            r = ok ? (n ? (e & 0x0FFFFFFFu) : (kRecRun | len)) : 0u;
        }
        return ok ? bits : 0u;
    };
    bool stuck = false; // (the lean form's fast steps: the token at pos is known not to be a simple one)
    while (pos < lim) {
        if (!Rich) {
            // FAST steps of the lean form, while the position is kFastRoom bits and more in front of the limit -- nearly all of a
            // subsequence: two simple tokens end in front of the limit whatever they are, so a step is two lookups, two adds and the
            // record's store, no question asked.  A lane whose next token is not a simple one leaves for the careful step below.
            // (Round 6 measured this loop twice: with the staging loop in front of it -- half of a workgroup's time without a vector
            //  instruction -- it bought 1 .. 4 %, profiles/r06x_sync_fastloop_ab.txt; behind the staging's fix the decode is what a
            //  workgroup does, profiles/r14_sync_fastloop_ab.txt.)
            for (;;) {
                if (!(pos + kFastRoom <= lim) || stuck) break;
                const uint32_t w = in.window(pos);
                uint32_t ea = lut[w & (kLutEntries - 1)];
                ea = ea >= kEntSimpleMin ? ea : 0u;
                const uint32_t ba = ea >> 28;
                uint32_t eb = lut[(w >> ba) & (kLutEntries - 1)]; // (ba == 0: the same entry again, masked again)
                eb = eb >= kEntSimpleMin ? eb : 0u;
                const uint32_t bb = eb >> 28;
                if (Count) {
                    bytes += ent_out_bytes(ea) + ent_out_bytes(eb);
                    rec.put2(ea & 0x0FFFFFFFu, eb & 0x0FFFFFFFu);
                }
                pos += ba + bb;
                stuck = bb == 0; // (the token now at pos is not a simple one)
            }
            if (!(pos < lim)) break;
        }
        const uint32_t w = in.window(pos), room = lim - pos;
        uint32_t ra = 0, rb = 0;
        const uint32_t ba = take(w, room, true, ra);
        const bool en_b = ba != 0 && ba <= 14 && ba < room; // (the second token starts inside, with 18 valid bits in the window)
        const uint32_t bb = take(w >> ba, room - ba, en_b, rb);
        if (Count) rec.put2(ra, rb);
        pos += ba + bb;
        const bool waiting = pos < lim && (ba == 0 || (en_b && bb == 0)); // the token at pos is not a plain one
        stuck = waiting;
        if (Vote::go(waiting)) {
            uint32_t n3, l3 = 0, run = 0, bits;
            const uint32_t kind = fetch(in.window(pos), lut, lenof, lim - pos, n3, l3, run, bits);
            if (kind >= kTokEob) {
                flags = kind == kTokEob ? kSubEob : kSubInvalid;
                if (Count) c.eob = pos + bits;
                break;
            }
            pos += bits;
            stuck = false;
            if (Count) {
                if (kind == kTokLit)
                    bytes += n3, rec.put2(n3 << 26 | l3, 0u);
                else
                    bytes += run, rec.put2(kRecRun | run, 0u);
            }
            c.gen += kind == kTokMatch ? 1u : 0u;
        }
    }
    if (!flags && pos < limit) flags = kSubInvalid; // ran off the data without an end-of-block symbol
    c.flags = flags;
    if (Count) c.bytes += bytes;
    return pos;
}

// a subsequence as the synchronisation keeps it
struct SubState {
    uint32_t start, end; // first bit of its first token; position behind its last one (its nominal boundary if it is flagged)
    SubCount c;
    uint32_t nrec;       // entries its decode left (more than kRecCap: kSubOverflow is set)
};
FPNG_DEC_HD void sub_close(SubState &s, uint32_t boundary, uint32_t e, uint32_t nrec)
{
    // A decode that derailed or met an end-of-block symbol hands over on the nominal boundary: speculative decodes meet FALSE
    // end-of-block symbols; which one is the true one is settled afterwards (the first one of the chain).
    s.end = s.c.flags ? boundary : e;
    s.nrec = nrec < 1023u ? nrec : 1023u;
    if (nrec > kRecCap) s.c.flags |= kSubOverflow;
}

// First decode of a subsequence [nominal, boundary): the decoder starts `lead` bits EARLIER and has, with a probability that
// tools/sync_stats.c measured (128 bits: all but 0.04 % of the subsequences of a synthetic gradient, 1.8 % of a photograph),
// fallen into step with the true token sequence when it crosses `nominal`; the first token boundary at or behind `nominal` is
// the subsequence's start.  Whether it is the true one shows when it is compared with the predecessor's end.
// sub_lead: the lead-in (the lean walk; gen: the matches it met that a rich walk would have taken in its stride) -> the start;
// sub_main: the subsequence's own tokens, recorded in rec, with the walk the caller chose.
template <class Vote, class Bits>
FPNG_DEC_HD uint32_t sub_lead(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t lead_start, uint32_t nominal, uint32_t data_limit, uint32_t &gen)
{
    // (in two halves: a decoder that is not in step yet decodes noise, matches of every kind among it -- what the second half meets
    //  says something about the stream)
    NoRec none;
    uint32_t p = lead_start;
    const uint32_t mid = nominal - lead_start > 64u ? nominal - 64u : lead_start;
    for (uint32_t half = 0; half < 2; half++) { // (one body of the walk for both halves)
        const uint32_t to = half ? nominal : mid;
        if (p >= to && !half) continue;
        SubCount d = {0, 0, 0, 0};
        p = walk_count<false, false, Vote>(in, lut, lenof, p, to, data_limit, d, none);
        if (d.flags || p < to) p = to; // (it derailed: any start is as good as another)
        gen = d.gen;
    }
    return p;
}
template <class Vote, class Bits, class Rec>
FPNG_DEC_HD void sub_main(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t start, uint32_t boundary, uint32_t data_limit, SubState &s, Rec &rec, bool rich)
{
    s.start = start;
    s.c.bytes = s.c.flags = s.c.eob = s.c.gen = 0;
    const uint32_t e = rich ? walk_count<true, true, Vote>(in, lut, lenof, start, boundary, data_limit, s.c, rec)
                            : walk_count<true, false, Vote>(in, lut, lenof, start, boundary, data_limit, s.c, rec);
    sub_close(s, boundary, e, rec.count());
}

// The subsequence must start at `want` instead of s.start (its predecessor ended there): decoded again as a whole, its records
// written again from the first one.  (Until round 5 the two decodes were stepped token by token to where they meet and only the
// counts in front of that point exchanged; records cannot be patched that way -- the new decode may need more of them in front of the
// meeting point than the old one left room for.)
template <class Vote, class Bits, class Rec>
FPNG_DEC_HD void sub_redo(const Bits &in, const uint32_t *lut, const uint8_t *lenof, uint32_t want, uint32_t boundary, uint32_t data_limit, SubState &s, Rec &rec, bool rich)
{
    sub_main<Vote>(in, lut, lenof, want, boundary, data_limit, s, rec, rich);
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
    NoRec none;
    const uint32_t e = walk_count<false, true, Vote>(in, lut, lenof, start, boundary, data_limit, d, none); // (the rich walk: periodic streams are streams of matches)
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
        NoRec none;
        const uint32_t p = walk_count<false, true, Vote>(in, lut, lenof, lead_start + j, nominal, data_limit, d, none);
        if (d.flags || p < nominal || pm_at(map, p - nominal) != kPhaseUnknown) continue;
        pm_set(map, p - nominal, sub_probe<Vote>(in, lut, lenof, p, boundary, data_limit) - boundary);
    }
}

// per-subsequence word in global memory: start - nominal (0..17) | end - boundary (0..17) << 5 | flags << 10 | entries (10 bits) << 13
FPNG_DEC_HD uint32_t pack_info(uint32_t start_rel, uint32_t end_rel, const SubCount &c, uint32_t nrec) { return start_rel | end_rel << 5 | c.flags << 10 | nrec << 13; }
FPNG_DEC_HD uint32_t info_start(uint32_t v) { return v & 31u; }
FPNG_DEC_HD uint32_t info_end(uint32_t v) { return (v >> 5) & 31u; }
FPNG_DEC_HD uint32_t info_flags(uint32_t v) { return (v >> 10) & 7u; }
FPNG_DEC_HD uint32_t info_nrec(uint32_t v) { return (v >> 13) & 1023u; }

// ---- the pass that writes: records into WINDOWS of the filtered stream ----
// The filtered stream = what the reference's decoder consumes row by row (src/fpng.cpp:2255-2262): h rows of 1 filter byte +
// w * c bytes.  It is never written to memory as a whole: the kernel that undoes the Up filter (dec_unfilter_kernel) works on tiles
// of 48 rows x one block of columns, and fills each tile's rows from the records of the subsequences that cover them.  A WINDOW =
// the bytes of one row that one tile holds: column block cb of row y covers the row's bytes [xw, xe), xw = cb ? 1 + cb * cbw : 0 (the
// first block also holds the row's filter byte), xe = min(1 + (cb + 1) * cbw, stride); a window never crosses a row end.
enum : uint32_t { kEmitBadStream = 2u, kEmitLeaveToCpu = 16u }; // (= kDecBadStream, kDecStalled of decode.h)
struct Window {
    uint64_t ws;   // stream offset of its first byte
    uint32_t wlen; // its bytes
    uint32_t xw;   // column (place in the row, 0 = the filter byte) of its first byte
};
FPNG_DEC_HD Window window_of(uint32_t y, uint32_t cb, uint32_t cbw, uint32_t stride)
{
    Window w;
    w.xw = cb ? 1u + cb * cbw : 0u;
    const uint32_t xe = 1u + (cb + 1u) * cbw < stride ? 1u + (cb + 1u) * cbw : stride;
    w.wlen = xe - w.xw;
    w.ws = (uint64_t)y * stride + w.xw;
    return w;
}
// Every window whose FIRST byte lies in the stream bytes [off, off + bytes) -- the output of one subsequence -- is handed to
// mark(y, cb): that subsequence is where the window's walk starts (dec_subscan_kernel writes the index the tiles read).
template <class Mark> FPNG_DEC_HD void for_windows_starting_in(uint64_t off, uint32_t bytes, uint32_t cbw, uint32_t ncb, uint32_t stride, uint32_t h, const Mark &mark)
{
    if (!bytes) return;
    uint32_t y = (off >> 32) ? (uint32_t)(off / stride) : (uint32_t)off / stride; // (a 32-bit division wherever the image is under 4 GB)
    const uint32_t x = (uint32_t)(off - (uint64_t)y * stride);
    // first window start at or behind column x of row y: 0, then 1 + cb * cbw
    uint32_t cb = x == 0 ? 0u : (x - 1u + cbw - 1u) / cbw;
    if (x != 0 && cb == 0) cb = 1; // (x = 1: the start "1 + 0 * cbw" is not a window start -- block 0 starts at the filter byte)
    const uint64_t end = off + bytes;
    for (; y < h; y++, cb = 0) {
        for (; cb < ncb; cb++) {
            const uint64_t ws = (uint64_t)y * stride + (cb ? 1u + cb * cbw : 0u);
            if (ws >= end) return;
            mark(y, cb);
        }
    }
}

// ---- the walk over a subsequence's records that fills one window (dec_unfilter_kernel's tiles; tests/cpp/decode_emul.cpp) ----
// EXACT STORES.  The walk keeps the last eight output bytes (the "tail": tl, th -- the most recent byte highest in th) and every
// step stores the eight bytes that END at the step's last byte: all of them true output, whatever the step added (0 .. 7 bytes), so no
// store ever leaves a byte that is not the stream's -- with two exceptions, both inside the walk's own territory: (a) a store never
// begins in front of the subsequence's first byte (c0: the bytes in front are another walk's); in the subsequence's first eight
// bytes it begins there, shifted, and leaves zeros on bytes that this walk's next steps write; (b) the bytes of a LONG match (more
// than eight bytes) are not written at all: the walk marks the match's pixels in the window's bitmap, and when every walk of the tile
// has ended the marked pixels take the value of the nearest unmarked pixel to their left (a match repeats the pixel in front of it:
// reference src/fpng.cpp:2273-2330) -- all pixels of all long matches at once, a lane each, instead of one lane writing 258 bytes.
// The tail's older bytes may then stand for bytes of such a match (stale): they land on marked pixels only.  So walks never write a
// byte of another walk's territory, and the order in which the lanes of a wave -- or two waves -- store does not matter.
// A pixel the window's FIRST data byte belongs to may be marked too (a match that began in the window to the left): the walk whose
// match covers that byte hands its pixel over (Out::entry_px), the source of the window's leading marked pixels.
// Out (decode.hip: TileOut; the emulator: HostOut):  put64(pos, v) -- eight bytes at window position pos in [-8, wlen] (eight bytes
// of slack on either side); mark(lo, hi) -- pixels [lo, hi) of the window are a long match's; entry_px(th) -- see above.
// Checked for the matches that START in this window (every match starts in exactly one): a match repeats whole pixels, starts on a
// pixel, stays inside its row (reference src/fpng.cpp:2273-2330); one at a row's FIRST pixel would repeat a pixel of zeros (:2268,
// prev_delta_* start at 0) -- "the last literal bytes" know nothing of rows, no fpng encoder writes such a match, the file is left
// to the CPU decoder.  That every row starts with its filter literal is checked where the rows are read (dec_unfilter_kernel).
struct WalkState {
    int32_t c, c0;   // window position of the next output byte; of the subsequence's first byte
    uint32_t tl, th; // the tail
    uint32_t err;    // kEmit* flags
};
FPNG_DEC_HD uint64_t shr64(uint64_t v, uint32_t s) { return v >> (s & 63u); }
FPNG_DEC_HD uint64_t shl64(uint64_t v, uint32_t s) { return v << (s & 63u); }
// n (0 .. 7) bytes d (first one lowest; n == 0: d == 0) are the walk's next output.  Hold = false: the caller knows that the walk is
// eight bytes and more into its subsequence (every entry holds a byte at least: behind a subsequence's first eight entries) -- the
// store begins eight bytes in front of the step's end, no question.
// Wide: n may be 8 (a match of two 4-byte pixels), the whole tail is new then.
template <bool Hold = true, bool Wide = false, class Out> FPNG_DEC_HD void walk_apply(WalkState &s, uint32_t n, uint64_t d, const Window &w, Out &out)
{
    const uint32_t sh = 8u * n;
    uint64_t ta = shr64((uint64_t)s.th << 32 | s.tl, sh) | shl64(d, 64u - sh);
    if (Wide) ta = n >= 8u ? d : ta;
    s.tl = (uint32_t)ta, s.th = (uint32_t)(ta >> 32);
    const int32_t e = s.c + (int32_t)n, e8 = e - 8, start = (Hold && s.c0 > e8) ? s.c0 : e8;
    const uint64_t v = Hold ? shr64(ta, 8u * (uint32_t)(start - e8)) : ta; // (start - e8 == 8: nothing of this subsequence yet -- whatever is stored lies where its first steps write)
    const int32_t lo = start < -8 ? -8 : start, pos = lo > (int32_t)w.wlen ? (int32_t)w.wlen : lo;
    out.put64(pos, v);
    s.c = e;
}
// the pixel a match repeats, and k copies of it (C * k <= 8) as output bytes
template <int C> FPNG_DEC_HD uint32_t tail_px(uint32_t th) { return C == 4 ? th : th >> 8; }
// Is the entry (a, b) one for the straight-line step -- each record nothing, a group of literals, or a match of ONE pixel, seven
// bytes at most together?
template <int C> FPNG_DEC_HD bool entry_plain(uint32_t a, uint32_t b)
{
    const bool ra = (a & kRecRun) != 0, rb = (b & kRecRun) != 0;
    const bool oka = !ra || (a & 0xFFFFFFu) == (uint32_t)C, okb = !rb || (b & 0xFFFFFFu) == (uint32_t)C;
    return oka && okb && !(C == 4 && ra && rb);
}
// ... the straight-line step for such an entry: both records become bytes (a match of one pixel: the tail's pixel), one store.
// The reference's checks for a match that begins in the window (see walk_record): a one-pixel match cannot leave its row if it
// starts on a pixel, so that is all there is to look at.
template <int C, bool Hold = true, class Out> FPNG_DEC_HD void walk_entry_plain(uint32_t a, uint32_t b, WalkState &s, const Window &w, uint32_t stride, Out &out)
{
    const bool ra = (a & kRecRun) != 0, rb = (b & kRecRun) != 0;
    const uint32_t na = ra ? (uint32_t)C : (a >> 26) & 3u, da = ra ? tail_px<C>(s.th) : a & 0xFFFFFFu;
    const uint32_t th1 = funnel(da, s.th, (8u * na) & 31u); // the tail's upper half behind a (four bytes from a 4-byte pixel: as it was)
    const uint32_t nb = rb ? (uint32_t)C : (b >> 26) & 3u, db = rb ? tail_px<C>(th1) : b & 0xFFFFFFu;
    if (ra | rb) {
        const uint32_t ca = (uint32_t)s.c, cb2 = ca + na, bpl = stride - 1; // (as unsigned numbers: "not in front of the window" too)
        uint32_t bad = 0;
        if (ra && ca < w.wlen) {
            const uint32_t rowleft = stride - (w.xw + ca);
            bad |= (rowleft % C != 0 || rowleft > bpl) ? kEmitBadStream : (rowleft == bpl ? kEmitLeaveToCpu : 0u);
        }
        if (rb && cb2 < w.wlen) {
            const uint32_t rowleft = stride - (w.xw + cb2);
            bad |= (rowleft % C != 0 || rowleft > bpl) ? kEmitBadStream : (rowleft == bpl ? kEmitLeaveToCpu : 0u);
        }
        s.err |= bad;
    }
    walk_apply<Hold>(s, na + nb, (uint64_t)da | shl64(db, 8u * na), w, out);
}
// ... and for an entry without a match (nearly all of them): two groups of literals, six bytes at most
template <bool Hold = true, class Out> FPNG_DEC_HD void walk_entry_literals(uint32_t a, uint32_t b, WalkState &s, const Window &w, Out &out)
{
    const uint32_t na = (a >> 26) & 3u, nb = (b >> 26) & 3u;
    walk_apply<Hold>(s, na + nb, (uint64_t)(a & 0xFFFFFFu) | shl64(b & 0xFFFFFFu, 8u * na), w, out);
}
// the reference's checks for a match of `run` bytes that begins at the walk's position, if that lies in this window
template <int C> FPNG_DEC_HD void match_checks(uint32_t run, WalkState &s, const Window &w, uint32_t stride)
{
    const uint32_t c = (uint32_t)s.c, bpl = stride - 1;
    if (c < w.wlen) {
        const uint32_t rowleft = stride - (w.xw + c); // bytes up to the end of the row, this one included (== stride: a filter byte stands here)
        if (rowleft % C != 0 || rowleft > bpl || run > rowleft || run % C != 0)
            s.err |= kEmitBadStream;
        else if (rowleft == bpl)
            s.err |= kEmitLeaveToCpu;
    }
}
// a LONG match (more than eight bytes -- or one no fpng encoder writes: the status says so): its pixels are marked.  fb: window
// position of the window's first DATA byte (1 where it begins with the row's filter byte).
template <int C, class Out> FPNG_DEC_HD void walk_long_match(uint32_t run, WalkState &s, const Window &w, uint32_t stride, Out &out)
{
    match_checks<C>(run, s, w, stride);
    const int32_t fb = w.xw ? 0 : 1, e = s.c + (int32_t)run;
    const int32_t lo = s.c > fb ? s.c : fb, hi = e < (int32_t)w.wlen ? e : (int32_t)w.wlen;
    if (lo < hi) {
        if (s.c <= fb) out.entry_px(s.th); // (it covers the window's first data byte)
        out.mark((uint32_t)(lo - fb) / C, ((uint32_t)(hi - fb) + C - 1) / C);
    }
    s.c = e;
}
// ... and one record in full
// (a group of literals and a match of one or two pixels -- output bytes like any others, the pixel once or twice -- share ONE store:
//  dithered panels are two-pixel matches between literals, entry after entry, in some lane of every wave)
template <int C, class Out> FPNG_DEC_HD void walk_record(uint32_t r, WalkState &s, const Window &w, uint32_t stride, Out &out)
{
    const bool isrun = (r & kRecRun) != 0;
    const uint32_t run = r & 0xFFFFFFu;
    if (isrun && !(run <= 8u && run % C == 0)) {
        walk_long_match<C>(run, s, w, stride, out);
        return;
    }
    if (isrun) match_checks<C>(run, s, w, stride);
    const uint64_t one = C == 4 ? tail_px<C>(s.th) : (tail_px<C>(s.th) & 0xFFFFFFu);
    const uint64_t rep = run > (uint32_t)C ? (one | one << (8 * C)) : one;
    const uint32_t n = isrun ? run : (r >> 26) & 3u;
    if (n) walk_apply<true, C == 4>(s, n, isrun ? rep : (uint64_t)(r & 0xFFFFFFu), w, out);
}
// Flat content: every step two matches of 258 bytes.  Is the entry one of long matches only (each record nothing or one)?
FPNG_DEC_HD bool entry_long_matches(uint32_t a, uint32_t b)
{
    const bool oka = !a || ((a & kRecRun) && (a & 0xFFFFFFu) > 8u), okb = !b || ((b & kRecRun) && (b & 0xFFFFFFu) > 8u);
    return oka && okb;
}
template <int C, class Out> FPNG_DEC_HD void walk_entry_long(uint32_t a, uint32_t b, WalkState &s, const Window &w, uint32_t stride, Out &out)
{
    if (a) walk_long_match<C>(a & 0xFFFFFFu, s, w, stride, out);
    if (b) walk_long_match<C>(b & 0xFFFFFFu, s, w, stride, out);
}
// Long matches BETWEEN literals (dithered panels, tiles whose runs end in a few literal pixels): each record nothing, a group of
// literals, a one-pixel match, or a long match.  The long ones are marked where they stand in the entry's order, what is left is the
// plain entry's one store.
FPNG_DEC_HD bool record_long_match(uint32_t r) { return (r & kRecRun) && (r & 0xFFFFFFu) > 8u; }
template <int C> FPNG_DEC_HD bool entry_mixed(uint32_t a, uint32_t b) { return entry_plain<C>(record_long_match(a) ? 0u : a, record_long_match(b) ? 0u : b); }
template <int C, class Out> FPNG_DEC_HD void walk_entry_mixed(uint32_t a, uint32_t b, WalkState &s, const Window &w, uint32_t stride, Out &out)
{
    const bool la = record_long_match(a), lb = record_long_match(b);
    if (la) walk_long_match<C>(a & 0xFFFFFFu, s, w, stride, out);
    walk_entry_plain<C, true>(la ? 0u : a, lb ? 0u : b, s, w, stride, out); // (nothing left: the tail's eight bytes once more -- true ones, or a long match's: marked)
    if (lb) walk_long_match<C>(b & 0xFFFFFFu, s, w, stride, out);
}
template <int C, class Out> FPNG_DEC_HD void walk_entry(uint64_t en, WalkState &s, const Window &w, uint32_t stride, Out &out)
{
    walk_record<C>((uint32_t)en, s, w, stride, out);
    walk_record<C>((uint32_t)(en >> 32), s, w, stride, out);
}

// ---- resume points.  A subsequence whose output covers many windows (long matches: flat content -- eleven windows of an 8K row
// and more) would be walked from its first record for each of them, a lane alone skipping forty entries.  dec_subscan_kernel,
// which sees every subsequence once, walks such a subsequence's records once and leaves each window that begins in its output the
// place where that window's walk may begin: the first entry that reaches into the window, where that entry's output begins
// (relative to the window's first byte: <= 0), the tail's upper half there.  A window's words (DecJob::win): subsequence, entry
// (kNoResume: from the subsequence's first record, its offset and tail as the synchronisation left them), position, tail. ----
constexpr uint32_t kWinWords = 4, kNoResume = 0xFFFFFFFFu;
constexpr uint32_t kResumeMinBytes = 2048; // output bytes of a subsequence from which on its windows get resume points (1024, 512: no different, profiles/r15_dither_notes.txt)
constexpr uint32_t kResumeAlign = 4;       // a resume point is an entry whose number is a multiple of this (the tiles read entries in batches of four)
struct ResumeWalk {
    uint32_t k, c, th;    // the next entry; output bytes of the entries in front of it; the tail's upper half there
    uint32_t sk, sc, sth; // ... the same at the last entry whose number is a multiple of kResumeAlign
};
FPNG_DEC_HD ResumeWalk resume_begin(uint32_t th) { return ResumeWalk{0u, 0u, th, 0u, 0u, th}; }
// the walk moves on to the first entry that ends behind byte d of the subsequence's output (d rises from call to call); the resume
// point for a window that begins at byte d: entry sk, whose output begins sc - d bytes from the window's first (<= 0), tail sth
template <class Entry> FPNG_DEC_HD void resume_seek(ResumeWalk &s, uint32_t d, uint32_t nent, const Entry &entry)
{
    for (; s.k < nent; s.k++) {
        if (!(s.k & (kResumeAlign - 1u))) s.sk = s.k, s.sc = s.c, s.sth = s.th;
        const uint64_t en = entry(s.k);
        uint32_t nbytes = 0, th = s.th;
        for (int half = 0; half < 2; half++) {
            const uint32_t r = (uint32_t)(en >> (32 * half));
            if (r & kRecRun)
                nbytes += r & 0xFFFFFFu; // (a match leaves the tail's pixel as it is)
            else {
                const uint32_t n = (r >> 26) & 3u;
                nbytes += n, th = funnel(r & 0xFFFFFFu, th, 8 * n);
            }
        }
        if (s.c + nbytes > d) break;
        s.c += nbytes, s.th = th;
    }
}

// Does a subsequence's walk ever look at the literal bytes in front of it?  Only if a match comes before four literal bytes of its
// own -- on photographic content next to never, and then the look back (below: loads all over the records) can be left out.
// e0, e1: its first two entries (the lanes of a wave read theirs side by side).
FPNG_DEC_HD bool needs_lastpx(uint64_t e0, uint64_t e1, uint32_t nent)
{
    uint32_t lits = 0;
    const uint32_t r[4] = {(uint32_t)e0, (uint32_t)(e0 >> 32), nent > 1 ? (uint32_t)e1 : 0u, nent > 1 ? (uint32_t)(e1 >> 32) : 0u};
    for (int q = 0; q < 4; q++) {
        if (r[q] & kRecRun) return lits < 4;
        lits += (r[q] >> 26) & 3u;
    }
    return lits < 4 && nent > 2; // (two entries at most and no match among them: nothing ever asks)
}

// The four literal bytes in front of subsequence g (what a match at its very beginning repeats): collected backwards over the
// records of the subsequences in front of it -- matches carry no bytes of their own, so usually the last two or three records of
// the subsequence just in front.  nent(k): entries of the file's subsequence k (at most kRecCap of them were kept); entry(k, e):
// its entry e as (first record) | (second record) << 32.
template <class Nent, class Entry> FPNG_DEC_HD uint32_t lookback_lastpx(uint32_t g, const Nent &nent, const Entry &entry)
{
    uint32_t v = 0, got = 0; // got bytes collected, the most recent one in bits 31..24
    while (g > 0 && got < 4) {
        g--;
        uint32_t ne = nent(g);
        ne = ne < kRecCap ? ne : kRecCap;
        while (ne > 0 && got < 4) {
            const uint64_t en = entry(g, --ne);
            for (int half = 1; half >= 0 && got < 4; half--) {
                const uint32_t r = (uint32_t)(en >> (32 * half)), n = (r >> 26) & 3u;
                if (!n) continue;
                const uint32_t m = n < 4 - got ? n : 4 - got;                                // take its m most recent bytes
                const uint32_t top = ((r & 0xFFFFFFu) << (8 * (4 - n))) >> (32 - 8 * m);      // (its n bytes top-aligned, the upper m of them)
                v |= top << (32 - 8 * (got + m));
                got += m;
            }
        }
    }
    return v;
}

} // namespace dec
} // namespace fpng_amd
