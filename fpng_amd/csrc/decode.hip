// decode.hip -- GPU batch decoder of fpng-written PNG files (reference src/fpng.cpp:2209-2901 decodes the same
// streams serially) in its data-parallel form: the container and the Deflate block header are parsed on the host
// (a few hundred bytes per file, decode_api.cpp); everything that touches the pixel stream runs here.
//
// An fpng stream is ONE Huffman-coded bit string with no restart points, but Huffman decoders SELF-SYNCHRONISE: started at a
// wrong bit, a decoder falls into step with the true token sequence after a few dozen bits.  So the token bits are cut into
// subsequences of kSubBits bits, one thread each (the per-thread logic lives in decode_core.h, which also compiles for the host),
// and EVERY TOKEN IS DECODED ONCE (round 6): the decode that settles a subsequence leaves token records, the pass that writes reads them.
//   dec_build_lut_kernel the lookup table of every distinct set of code lengths (decode_core.h: up to three literals per lookup)
//   dec_sync_kernel     round 0 (<false>), one workgroup per kDecSubBlock subsequences, their bits and the lookup table staged in
//                       LDS: every thread starts kDecLeadIn bits EARLY, takes the first token boundary at or behind its nominal
//                       first bit as its start, decodes its tokens -- their output bytes counted, what they are left behind as
//                       records (an 8-byte entry per step of the walk) -- then, still inside the workgroup, every thread whose
//                       predecessor ended elsewhere than it started is corrected -- a few where they are, from four on gathered
//                       into one wave, decoded again, their records written again -- until nothing changes, three steps at most:
//                       a workgroup that still changes then (a periodic stream) is marked and left to round 1.  Rounds 1..
//                       (<true>, persistent workgroups): the same across workgroup borders -- a workgroup whose first thread
//                       starts where the previous workgroup's last one ended is passed over -- plus the PHASE MAPS that settle
//                       periodic streams.
//   dec_chain_kernel    in front of rounds 2..: the workgroups' phase maps composed along every file -> the entry each workgroup
//                       must take (leaves at once unless some map has more than one pair)
//   dec_offsets_kernel  per file: the first end-of-block symbol of the chain ends the stream; up to there the chain must hold and
//                       no subsequence may be invalid; exclusive scan of the workgroups' byte counts; the total must be the image
//   dec_subscan_kernel  per workgroup of subsequences: output offset of every subsequence, the four literal bytes in front of it
//                       (what a match at its very beginning repeats), and for every WINDOW (a row x 256 pixels: a row of a tile of
//                       the pass that writes) that begins in a subsequence's output: the subsequence, and -- flat content -- where in
//                       its records the window's walk may begin
//   dec_unfilter_kernel the pass that writes, one workgroup per tile of kDecUnfRows rows x 256 pixels: the tile's rows filled in LDS
//                       from the records (a walk per (window, subsequence) pair, dealt one per thread; exact stores; the pixels of
//                       long matches marked and filled a lane each) -- every rule of the reference's decoder checked (filter
//                       literal 0 then 2, matches whole pixels inside a row; that the stream ends 4 bytes before the IDAT does:
//                       dec_offsets_kernel) -- then the Up filter undone in the same pass: a thread per dword column, the sums of
//                       the segments above through a decoupled look-back; channel count conversion
//   dec_stored_kernel   files that are stored blocks (reference fpng.cpp:2107-2207): a strided copy
#include "decode.h"
#include "decode_core.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <type_traits>
#include <vector>

namespace fpng_amd {

namespace {

using namespace dec;

constexpr int kWave = 64;
constexpr int kDecBlock = 256;
constexpr int kSubBlock = (int)kDecSubBlock;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifdef FPNG_DEC_SYNC_TIMING // diagnostic build (fpng_amd/build.py --variant sync_timing): where does a workgroup of round 0 spend its time?
__device__ unsigned long long g_sync_times[8 * 65536];
#define FPNG_SYNC_STAMP(k) do { if (!CAND && threadIdx.x == 0 && blockIdx.x < 65536) g_sync_times[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define FPNG_SYNC_STAMP(k) do { } while (0)
#endif

// The token bits a workgroup works on are staged in LDS first (coalesced loads): thread t decodes the kSubBits bits that start
// kSubBits / 32 dwords behind thread t-1's, so straight from global memory every load instruction of a wave would touch 64
// different cache lines.  In LDS every 32 dwords of the slice are followed by a COPY of the next dword: dword d sits at d + d / 32
// and its successor always in the next slot (one ds_read2 per window), and the lanes of a wave, 16 dwords apart, meet in
// different banks.
__device__ __forceinline__ uint32_t slice_slot(uint32_t d) { return d + (d >> 5); }
constexpr uint32_t slice_slots(uint32_t dwords) { return dwords + dwords / 32 + 2; }

struct LdsBits {
    const uint32_t *l;
    __device__ __forceinline__ uint32_t window(uint32_t pos) const
    {
        // (the dword index hidden from the optimiser, which otherwise spreads the "* 4" of the address over both shifts: five vector
        //  instructions for the slot's byte address instead of shift, shift, add-shift)
        uint32_t d = pos >> 5;
        asm("" : "+v"(d));
        const uint32_t s = d + (d >> 5);
        return __builtin_amdgcn_alignbit(l[s + 1], l[s], pos & 31u);
    }
};

struct SyncJob { // what dec_sync_kernel reads of a file's record
    uint64_t first_bit, end_limit_bit, z_bytes;
    const uint8_t *z;
    const uint32_t *lut;
    uint32_t n_sub;
};
__device__ __forceinline__ SyncJob sync_job_of(const DecJob &j)
{
    SyncJob r;
    r.first_bit = j.first_bit, r.end_limit_bit = j.end_limit_bit, r.z_bytes = j.z_bytes, r.z = j.z, r.lut = j.lut, r.n_sub = j.n_sub;
    return r;
}
// (Both stagings ask for EVERYTHING first and store afterwards: a loop of load -> wait -> store is a chain of round trips to memory --
//  seventeen of them for the bits, 22 us of a workgroup's 50, tools/gpu_sync_times.sh -- where one, or three, will do.)
template <int THREADS, int FLIGHT> __device__ __forceinline__ void stage_lut(const SyncJob &job, uint32_t *lut)
{
    const u32x4 *src = (const u32x4 *)job.lut;
    constexpr int kVec = (int)(kLutDwords / 4), kPer = (kVec + THREADS - 1) / THREADS, kFlight = FLIGHT; // (vectors in flight per thread: what the kernel's registers allow)
#pragma unroll 1
    for (int q0 = 0; q0 < kPer; q0 += kFlight) {
        u32x4 v[kFlight];
#pragma unroll
        for (int q = 0; q < kFlight; q++) {
            const int i = (int)threadIdx.x + (q0 + q) * THREADS;
            v[q] = src[i < kVec ? i : kVec - 1];
        }
#pragma unroll
        for (int q = 0; q < kFlight; q++) {
            const int i = (int)threadIdx.x + (q0 + q) * THREADS;
            if (i < kVec) ((u32x4 *)lut)[i] = v[q];
        }
    }
}
// dwords [d0, d0 + COUNT) of the file's zlib stream
template <int THREADS, uint32_t COUNT> __device__ __forceinline__ void stage_bits(const SyncJob &job, uint64_t d0, uint32_t *bits)
{
    const uint64_t n_dw = (job.z_bytes + 16) >> 2;
    const uint32_t *w = (const uint32_t *)job.z;
    constexpr uint32_t kPer = (COUNT + THREADS - 1) / THREADS, kFlight = 5; // (loads in flight per thread: the kernel has few registers to spare)
#pragma unroll 1
    for (uint32_t q0 = 0; q0 < kPer; q0 += kFlight) {
        uint32_t v[kFlight];
#pragma unroll
        for (uint32_t q = 0; q < kFlight; q++) {
            const uint32_t d = threadIdx.x + (q0 + q) * THREADS;
            const bool in = d < COUNT && d0 + d < n_dw;
            v[q] = w[in ? d0 + d : 0];
            v[q] = in ? v[q] : 0u;
        }
#pragma unroll
        for (uint32_t q = 0; q < kFlight; q++) {
            const uint32_t d = threadIdx.x + (q0 + q) * THREADS;
            if (d < COUNT) {
                const uint32_t sl = slice_slot(d);
                bits[sl] = v[q];
                if (d && !(d & 31u)) bits[sl - 1] = v[q]; // the copy behind the 32 dwords in front
            }
        }
    }
}

__device__ __forceinline__ const DecJob &job_of_sub(const DecJob *jobs, uint32_t n_jobs, uint32_t g, uint32_t &local)
{
    uint32_t lo = 0, hi = n_jobs; // (jobs are few; stored files have no subsequences and share their successor's base)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs[mid].sub_base <= g) lo = mid; else hi = mid;
    }
    local = g - jobs[lo].sub_base;
    return jobs[lo];
}
// ... the same by the lanes of a wave at once, and what dec_sync_kernel wants of the file's record with it (a batch has few files; the
// binary search and then the record's fields are a chain of round trips to memory in front of everything a workgroup does: here one)
__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t lane) { return (uint64_t)rl32((uint32_t)(v >> 32), lane) << 32 | rl32((uint32_t)v, lane); }
__device__ __forceinline__ SyncJob job_of_sub_wave(const DecJob *jobs, uint32_t n_jobs, uint32_t g, uint32_t &local)
{
    if (n_jobs > (uint32_t)kWave) return sync_job_of(job_of_sub(jobs, n_jobs, g, local));
    const uint32_t l = threadIdx.x & (kWave - 1);
    const DecJob &mine = jobs[l < n_jobs ? l : 0u];
    const uint32_t base = mine.sub_base, n_sub = mine.n_sub;
    const uint64_t first_bit = mine.first_bit, end_limit_bit = mine.end_limit_bit, z_bytes = mine.z_bytes, z = (uint64_t)(uintptr_t)mine.z, lut = (uint64_t)(uintptr_t)mine.lut;
    const uint32_t lo = (uint32_t)__popcll(__ballot(l < n_jobs && base <= g)) - 1u; // (bases rise, the first one is 0)
    local = g - rl32(base, lo);
    SyncJob r;
    r.n_sub = rl32(n_sub, lo), r.first_bit = rl64(first_bit, lo), r.end_limit_bit = rl64(end_limit_bit, lo), r.z_bytes = rl64(z_bytes, lo);
    r.z = (const uint8_t *)(uintptr_t)rl64(z, lo), r.lut = (const uint32_t *)(uintptr_t)rl64(lut, lo);
    return r;
}
template <int WAVES> __device__ __forceinline__ uint32_t block_min(uint32_t v, uint32_t *red) // red: LDS, WAVES words
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, kWave));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = red[0];
#pragma unroll
    for (int q = 1; q < WAVES; q++) r = min(r, red[q]);
    return r;
}
template <int WAVES> __device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t *red)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, kWave);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t r = red[0];
#pragma unroll
    for (int q = 1; q < WAVES; q++) r += red[q];
    return r;
}

#ifndef FPNG_DEC_PERSISTENT // 1: a few workgroups per compute unit loop over the blocks; 0: one workgroup per block
#define FPNG_DEC_PERSISTENT 0
#endif

// ---- token records (decode_core.h): where a subsequence's settling decode leaves what it decoded, an 8-byte entry per step of the
//      walk.  The 64 lanes of a wave write side by side -- entry k of lane l at 8-byte word k * 64 + l of the wave's chunk -- and
//      walk in step, so their stores fill whole lines.  A step that decoded nothing plain leaves its (empty) entry where the next
//      step's will go. ----
typedef __attribute__((address_space(1))) uint64_t gu64e;
struct TokOut {
    gu64e *col; // the subsequence's entry 0 (decode_core.h: rec_index)
    uint32_t k = 0;
    __device__ __forceinline__ void put2(uint32_t a, uint32_t b)
    {
        const uint32_t kk = k < kRecCap ? k : kRecCap;
        col[(kk >> 2) * 256u + (kk & 3u) * 8u] = (uint64_t)b << 32 | a;
        k += (a | b) ? 1u : 0u;
    }
    __device__ __forceinline__ uint32_t count() const { return k; }
};
__device__ __forceinline__ TokOut tok_of(uint64_t *tok, uint32_t g)
{
    TokOut o;
    o.col = (gu64e *)(uintptr_t)(tok + rec_index(g, 0));
    return o;
}

// ---- phase maps (decode_core.h) across the threads of a workgroup ----
__device__ __forceinline__ PhaseMap pm_shfl_up(const PhaseMap &v, int o)
{
    PhaseMap r;
    r.w[0] = (uint32_t)__shfl_up((int)v.w[0], o, kWave), r.w[1] = (uint32_t)__shfl_up((int)v.w[1], o, kWave), r.w[2] = (uint32_t)__shfl_up((int)v.w[2], o, kWave);
    return r;
}
// the map of thread t - 1 (one that knows nothing for thread 0); wtail: LDS, one entry per wave
__device__ __forceinline__ PhaseMap pm_of_prev_thread(const PhaseMap &v, PhaseMap *wtail)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    PhaseMap r = pm_shfl_up(v, 1);
    __syncthreads();
    if (lane == 63) wtail[wave] = v;
    __syncthreads();
    if (lane == 0) r = wave ? wtail[wave - 1] : pm_none();
    return r;
}
// inclusive prefix composition: thread t gets what the subsequences 0..t do to the phases thread 0's map knows
__device__ __forceinline__ PhaseMap pm_scan(PhaseMap g, PhaseMap *wtail)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
    for (int o = 1; o < kWave; o <<= 1) {
        const PhaseMap p = pm_shfl_up(g, o);
        if (lane >= (uint32_t)o) g = pm_compose(p, g);
    }
    __syncthreads();
    if (lane == 63) wtail[wave] = g;
    __syncthreads();
    if (wave) {
        PhaseMap pre = wtail[0];
        for (uint32_t q = 1; q < wave; q++) pre = pm_compose(pre, wtail[q]);
        g = pm_compose(pre, g);
    }
    return g;
}
__device__ __forceinline__ PhaseMap rec_map(const DecBlockRec &r)
{
    PhaseMap m;
    m.w[0] = r.map[0], m.w[1] = r.map[1], m.w[2] = r.map[2];
    return m;
}

// ---- synchronisation ----
constexpr uint32_t kGatherMin = 4; // threads of a workgroup to correct from which on they are gathered into one wave (dec_sync_kernel<true>)
enum : uint32_t { kLeftMany = 1, kLeftCrawling = 2 }; // DecBlockRec::left: why round 0 left the workgroup unsettled
#ifndef FPNG_DEC_PREFILTER // 0: the border rounds walk over their blocks one check after the other (A/B builds)
#define FPNG_DEC_PREFILTER 1
#endif
#ifndef FPNG_DEC_PAD_LDS
#define FPNG_DEC_PAD_LDS 0
#endif
// (keeps the probe's padding alive: the compiler cannot tell that the condition in front of it never holds)
__device__ __noinline__ void status_touch(uint32_t *p) { asm volatile("" ::"v"(p[0]) : "memory"); }
constexpr uint32_t kSyncDwords = kSubBlock * (kSubBits / 32) + kDecLeadIn / 32 + 1 + 3;

// CAND = false: round 0, the kernel nearly every subsequence of nearly every file is settled by -- kept lean: a workgroup whose
// corrections have not ended after kRefixRounds steps (a periodic stream: decode_core.h) is left as it is, marked (an EMPTY map
// in its record), and taken up by round 1.  CAND = true: the border rounds -- few workgroups have anything to do in them -- also
// know the candidate lists.
template <bool CAND>
__global__ __launch_bounds__(kSubBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void dec_sync_kernel(const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, uint32_t round,
                                                             DecSubArrays a, DecBlockRec *recs, uint32_t *changed, uint32_t *multi)
{
    __shared__ __attribute__((aligned(16))) uint32_t lut[kLutDwords];
    __shared__ uint32_t bits[slice_slots(kSyncDwords)];
    __shared__ uint32_t s_end[kSubBlock];
    __shared__ uint32_t red3[4 * (kSubBlock / kWave)];
    __shared__ PhaseMap wtail[kSubBlock / kWave + 1];
#if FPNG_DEC_PAD_LDS // occupancy probe (build variant): LDS nobody uses, so that fewer workgroups share a compute unit
    __shared__ uint32_t pad_lds[FPNG_DEC_PAD_LDS / 4];
    if (total_subs == 0xFFFFFFFFu) pad_lds[threadIdx.x] = round, status_touch(pad_lds);
#endif
    FPNG_SYNC_STAMP(0);
    const uint8_t *lenof = (const uint8_t *)(lut + kLutEntries);
    const uint32_t *staged = nullptr; // the table in LDS (workgroups are persistent: 1-pass files of one channel count share theirs)
    const uint32_t t = threadIdx.x;
    // Border rounds: nearly every block's border holds, and finding that out costs a chain of dependent loads per block (its file,
    // its record, its neighbour's) -- a persistent workgroup walking over its ten blocks one after the other spent 47 us per round on
    // nothing else.  The threads look at the workgroup's first kSubBlock blocks AT ONCE (thread t at the t-th); the walk below then
    // stops only at the blocks that are open.  (A record that another workgroup changes meanwhile is seen a round later, as before.)
    // (round 2 is launched blind behind round 1: when round 1 rewrote no record there is nothing for it to find)
    if (CAND && round == 2 && !*changed) return;
    __shared__ uint32_t open_bits[CAND ? kSubBlock / 32 : 1];
    if (CAND && round && FPNG_DEC_PREFILTER) {
        const uint32_t bi = blockIdx.x + t * gridDim.x;
        bool open = false;
        if (bi < n_blocks && (first_block + bi) * kSubBlock < total_subs) {
            const uint32_t blk = first_block + bi;
            uint32_t local0;
            (void)job_of_sub(jobs, n_jobs, blk * kSubBlock, local0);
            const DecBlockRec mine = recs[blk];
            const uint32_t prev = !local0 ? mine.entry_rel : (mine.want_rel != kDecWantUnknown ? mine.want_rel : recs[blk - 1].exit_rel);
            open = !(mine.entry_rel == prev && pm_count(rec_map(mine)));
        }
        const uint64_t m = __ballot(open);
        if ((t & 63) == 0) open_bits[(t >> 6) * 2] = (uint32_t)m, open_bits[(t >> 6) * 2 + 1] = (uint32_t)(m >> 32);
        __syncthreads();
    }
    for (uint32_t bi = blockIdx.x; bi < n_blocks; bi += gridDim.x) {
        if (CAND && round && FPNG_DEC_PREFILTER) {
            const uint32_t k = (bi - blockIdx.x) / gridDim.x;
            if (k < (uint32_t)kSubBlock && !((open_bits[k >> 5] >> (k & 31u)) & 1u)) continue; // (blocks behind the kSubBlock-th are looked at the old way)
        }
        const uint32_t blk = first_block + bi, g0 = blk * kSubBlock;
        if (g0 >= total_subs) break;
        // all subsequences of a workgroup's block belong to one file (sub_base is padded to kSubBlock by the host)
        uint32_t local0;
        const SyncJob job = CAND ? sync_job_of(job_of_sub(jobs, n_jobs, g0, local0)) : job_of_sub_wave(jobs, n_jobs, g0, local0); // (round 0, where every workgroup asks: fetched with the search)
        const uint32_t g = g0 + t, i = local0 + t;
        const bool valid = i < job.n_sub;
        uint32_t want0 = 0;
        bool cand_first = false;
        if (round) { // only the border to the previous block is open
            const DecBlockRec mine = recs[blk];
            const uint32_t pairs = pm_count(rec_map(mine));
            // (the file's first subsequence starts at the stream's first token: no border in front of the first block)
            const uint32_t prev = !local0 ? mine.entry_rel : (mine.want_rel != kDecWantUnknown ? mine.want_rel : recs[blk - 1].exit_rel); // (dec_chain_kernel's word, else the neighbour's)
            if (mine.entry_rel == prev && pairs) continue;
            want0 = prev;
            cand_first = pairs > 1 || (!pairs && mine.left == kLeftCrawling); // its corrections did not end in round 0, or it needed its phase maps before: straight to them
        }
        const uint32_t lead0 = local0 ? kDecLeadIn : 0u;
        const uint64_t first_nominal = job.first_bit + (uint64_t)local0 * kSubBits;
        const uint64_t d0 = (first_nominal - lead0) >> 5, base = d0 << 5;
        __syncthreads(); // (the previous block's LDS is free)
        if (staged != job.lut) stage_lut<kSubBlock, CAND ? 2 : 3>(job, lut), staged = job.lut;
        stage_bits<kSubBlock, kSyncDwords>(job, d0, bits);
        __syncthreads();
        FPNG_SYNC_STAMP(1);
        LdsBits in = {bits};
        const uint32_t nominal = (uint32_t)(first_nominal - base) + t * kSubBits, boundary = nominal + kSubBits;
        const uint64_t lim64 = job.end_limit_bit - base;
        const uint32_t data_limit = lim64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)lim64;
        SubState st;
        st.start = st.end = nominal;
        st.c.bytes = st.c.flags = st.c.eob = st.c.gen = 0, st.nrec = 0;
        bool dirty = false;
        // Which form of the walk (decode_core.h: walk_count)?  The first decode: the lean one unless the lead-ins of four lanes of the
        // wave and more met a match with extra bits -- then the stream is one of matches (UI content, tiles, flat rows), and the rich
        // one.  Every correction, here and in the border rounds, takes the rich one: few subsequences of a gradient are corrected,
        // nearly all of a periodic stream of matches are, again and again.
        constexpr bool kRichRedo = true;
        if (!round) {
            uint32_t p0 = nominal, gen = 0;
            if (valid && i) p0 = sub_lead<VoteAlone>(in, lut, lenof, nominal - kDecLeadIn, nominal, data_limit, gen);
            const bool rich = __popcll(__ballot(valid && gen != 0)) >= 4;
            if (valid) {
                TokOut rec = tok_of(a.tok, g);
                sub_main<VoteAlone>(in, lut, lenof, p0, boundary, data_limit, st, rec, rich);
                dirty = true;
            }
        }
        if (valid && round) {
            const uint32_t v = a.info[g];
            st.start = nominal + info_start(v), st.end = boundary + info_end(v), st.nrec = info_nrec(v);
            st.c.bytes = a.bytes[g], st.c.flags = info_flags(v);
            st.c.eob = (st.c.flags & kSubEob) ? nominal + a.eob[g] : 0u;
        }
        FPNG_SYNC_STAMP(2);
        want0 += nominal; // (thread 0's nominal: the block's first)
        s_end[t] = st.end;
        bool cand_done = false;
        uint32_t unsettled = 0;
        PhaseMap bmap = pm_none();
        for (uint32_t it = 0;; it++) {
            __syncthreads();
            if (it == 0) FPNG_SYNC_STAMP(3);
            uint32_t want = st.start;
            if (t)
                want = s_end[t - 1];
            else if (round)
                want = want0;
            const bool need = valid && want != st.start;
            const uint32_t n_need = (uint32_t)__syncthreads_count(need);
            if (!n_need) break;
            // round 0: a workgroup whose corrections do not end is left as it is, marked, and taken up by round 1, whose kernel knows the phase maps
            if (!CAND && it >= kRefixRounds) {
                unsettled = kLeftCrawling;
                break;
            }
            const bool cand_now = CAND && !cand_done && (it >= kRefixRounds || (cand_first && it >= 1)); // (thread 0 takes its wanted start in step 0)
            if (!cand_now) {
                // The threads that must be corrected: 0.04 % of a gradient's subsequences (one in every fifth workgroup: decoded again where
                // it is, its records written again), 2 % of a photograph's, more on flat content -- some lane
                // of nearly every wave then, and eight waves each wait for a lane or two.  From kGatherMin of them on they are GATHERED:
                // their slots are numbered over the workgroup, wave 0 decodes 64 of them per pass, the owners take the results back.
                // (Measured, 8 x 8K grad / 8 x 11 MP photograph / 8 x 8K stripes, sync ms: in place 0.80 / 0.81 / 0.25; all gathered
                //  and decoded again 1.01 / 0.65 / 0.13; all gathered and stepped until they meet 1.02 / 1.06 / 0.40.)
                if (n_need < kGatherMin) {
                    if (need) {
                        TokOut rec = tok_of(a.tok, g);
                        sub_redo<VoteAlone>(in, lut, lenof, want, boundary, data_limit, st, rec, kRichRedo);
                        s_end[t] = st.end;
                        dirty = true;
                    }
                    continue;
                }
                // (the exchange lives in s_end -- every thread holds its own end in a register and writes it back afterwards: a kilobyte
                //  more LDS would cost round 0's kernel its third workgroup per compute unit)
                uint32_t(*redo)[4] = (uint32_t(*)[4])s_end;
                uint32_t *redo_cnt = s_end + 4 * kWave;
                const uint64_t nm = __ballot(need);
                const uint32_t lane = t & 63, wv = t >> 6;
                if (lane == 0) redo_cnt[wv] = (uint32_t)__popcll(nm);
                __syncthreads();
                uint32_t slot = (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
                const uint32_t total = n_need;
#pragma unroll
                for (int q = 0; q < kSubBlock / kWave; q++) slot += (uint32_t)q < wv ? redo_cnt[q] : 0u;
                for (uint32_t base_slot = 0; base_slot < total; base_slot += kWave) { // (uniform)
                    const bool mine = need && slot >= base_slot && slot < base_slot + kWave;
                    if (mine) redo[slot - base_slot][0] = t << 5 | (want - nominal);
                    __syncthreads();
                    if (t < min(total - base_slot, (uint32_t)kWave)) {
                        uint32_t *q = redo[t];
                        const uint32_t v = q[0], ot = v >> 5, o_nominal = nominal - t * kSubBits + ot * kSubBits; // (this thread's nominal -> the owner's)
                        SubState r;
                        TokOut rec = tok_of(a.tok, g - t + ot); // (the owner's column: lanes of this pass write to columns of any wave's chunk)
                        sub_redo<VoteAlone>(in, lut, lenof, o_nominal + (v & 31u), o_nominal + kSubBits, data_limit, r, rec, kRichRedo);
                        q[0] = r.c.eob - o_nominal, q[1] = pack_info(v & 31u, r.end - (o_nominal + kSubBits), r.c, r.nrec), q[2] = r.c.bytes;
                    }
                    __syncthreads();
                    if (mine) {
                        const uint32_t *q = redo[slot - base_slot];
                        const uint32_t v = q[1];
                        st.start = want, st.end = boundary + info_end(v), st.nrec = info_nrec(v);
                        st.c.bytes = q[2], st.c.flags = info_flags(v), st.c.eob = nominal + q[0];
                        dirty = true;
                    }
                }
                __syncthreads();
                s_end[t] = st.end;
                continue;
            }
            // ---- the corrections crawl (a periodic stream): phase maps, decode_core.h ----
            if constexpr (CAND) {
                cand_done = true;
                PhaseMap map = valid ? pm_one(st.start - nominal, st.end - boundary) : pm_none();
                if (valid && i) pm_seed<VoteAlone>(in, lut, lenof, nominal - kDecLeadIn, nominal, boundary, data_limit, map);
                for (uint32_t step = 0; step < kCandGrowSteps; step++) { // (as many steps as there are phases the seeds missed; a map that stays incomplete costs its threads the old walk)
                    const PhaseMap pred = pm_of_prev_thread(map, wtail);
                    const bool grew = valid && t && pm_grow<VoteAlone>(in, lut, lenof, pred, nominal, boundary, data_limit, map);
                    if (!__syncthreads_or(grew)) break;
                }
                const PhaseMap g = pm_scan(map, wtail);
                const PhaseMap gp = pm_of_prev_thread(g, wtail);
                if (t == 0) wtail[kSubBlock / kWave].w[0] = st.start - nominal;
                __syncthreads();
                const uint32_t start0 = wtail[kSubBlock / kWave].w[0];
                const uint32_t srel = (valid && t) ? pm_at(gp, start0) : kPhaseUnknown;
                if (srel != kPhaseUnknown && nominal + srel != st.start) {
                    TokOut rec = tok_of(a.tok, g0 + t); // (`g` is the composed map here)
                    sub_redo<VoteAlone>(in, lut, lenof, nominal + srel, boundary, data_limit, st, rec, kRichRedo);
                    dirty = true;
                }
                __syncthreads();
                if (t == kSubBlock - 1) wtail[kSubBlock / kWave] = valid ? g : pm_none(); // (a block with fewer subsequences is its file's last)
                s_end[t] = st.end;
                __syncthreads();
                bmap = wtail[kSubBlock / kWave];
            }
        }
        FPNG_SYNC_STAMP(4);
        if (dirty && valid) {
            a.info[g] = pack_info(st.start - nominal, st.end - boundary, st.c, st.nrec);
            a.bytes[g] = st.c.bytes;
            if (st.c.flags & kSubEob) a.eob[g] = st.c.eob - nominal;
        }
        const bool any_dirty = __syncthreads_or(dirty);
        if (!any_dirty) continue;
        // the block's byte count and its first end-of-block / invalid subsequences: wave reductions, then ONE meeting in LDS
        uint32_t sum = valid ? st.c.bytes : 0u;
        uint32_t e = (valid && (st.c.flags & kSubEob)) ? t : (uint32_t)kSubBlock, inv = (valid && (st.c.flags & kSubInvalid)) ? t : (uint32_t)kSubBlock;
        uint32_t ovf = (valid && (st.c.flags & kSubOverflow)) ? t : (uint32_t)kSubBlock;
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            sum += (uint32_t)__shfl_xor((int)sum, o, kWave);
            e = min(e, (uint32_t)__shfl_xor((int)e, o, kWave));
            inv = min(inv, (uint32_t)__shfl_xor((int)inv, o, kWave));
            ovf = min(ovf, (uint32_t)__shfl_xor((int)ovf, o, kWave));
        }
        if ((t & 63) == 0) red3[(t >> 6) * 4] = sum, red3[(t >> 6) * 4 + 1] = e, red3[(t >> 6) * 4 + 2] = inv, red3[(t >> 6) * 4 + 3] = ovf;
        __syncthreads();
        if (t == 0) {
            sum = 0, e = inv = ovf = (uint32_t)kSubBlock;
            for (int q = 0; q < kSubBlock / kWave; q++) sum += red3[q * 4], e = min(e, red3[q * 4 + 1]), inv = min(inv, red3[q * 4 + 2]), ovf = min(ovf, red3[q * 4 + 3]);
            DecBlockRec r;
            r.sum = sum, r.first_eob = e, r.first_invalid = inv, r.first_overflow = ovf;
            r.entry_rel = st.start - nominal;
            r.exit_rel = s_end[kSubBlock - 1] - (nominal + (uint32_t)kSubBlock * kSubBits); // (a block with fewer subsequences is its file's last)
            const uint32_t known = pm_count(bmap);
            if (!known && !unsettled) bmap = pm_one(r.entry_rel, r.exit_rel & 31u);
            r.map[0] = bmap.w[0], r.map[1] = bmap.w[1], r.map[2] = bmap.w[2], r.want_rel = kDecWantUnknown, r.left = unsettled;
            recs[blk] = r;
            if (round) atomicOr(changed, 1u);
            if (CAND && known > 1) atomicOr(multi, 1u); // (dec_chain_kernel has something to do from now on)
        }
        FPNG_SYNC_STAMP(5);
    }
}

// ---- dec_chain_kernel, one workgroup per file, in front of every border round: where must every workgroup of subsequences be
//      entered?  Each one left a map (entry -> exit: one pair, or -- a periodic stream, decode_core.h -- one per phase); a workgroup
//      that is entered in a phase its map does not know hands on its present exit (its neighbour then asks it again next round, as
//      before).  The maps, completed that way to functions on the 18 possible phases, are composed along the file: a prefix "sum",
//      thread by thread over chunks of workgroups, Hillis-Steele across the threads.  Blocks [first_block, first_block + n_blocks)
//      of the batch: whole files, or (a file that arrives in pieces) a range of one file whose earlier blocks are settled. ----
__device__ __forceinline__ uint32_t block_fn(const DecBlockRec &r, uint32_t x)
{
    const uint32_t e = pm_at(rec_map(r), x);
    return e != kPhaseUnknown ? e : (r.exit_rel & 31u);
}
__global__ __launch_bounds__(kDecBlock) void dec_chain_kernel(const DecJob *jobs, uint32_t first_block, uint32_t n_blocks, DecBlockRec *recs, const uint32_t *multi)
{
    __shared__ uint32_t fw[2][kDecBlock][3];
    if (!*multi) return; // every workgroup's map is one pair: the neighbour's word is all there is to know (nearly every call ends here)
    const DecJob &job = jobs[blockIdx.x];
    if (job.mode != 0) return;
    const uint32_t t = threadIdx.x, fb0 = job.sub_base / kSubBlock, fnb = (job.n_sub + kSubBlock - 1) / kSubBlock;
    const uint32_t ra = max(fb0, first_block), rb = min(fb0 + fnb, first_block + n_blocks);
    if (ra >= rb) return;
    const uint32_t nb = rb - ra, per = (nb + kDecBlock - 1) / kDecBlock, i0 = min(t * per, nb), i1 = min(i0 + per, nb);
    PhaseMap f = pm_none(); // what this thread's chunk of workgroups does to a phase (a function on all of them here)
    for (uint32_t x = 0; x < kPhases; x++) pm_set(f, x, x);
    for (uint32_t b = i0; b < i1; b++) {
        const DecBlockRec r = recs[ra + b];
        PhaseMap n2 = f;
        for (uint32_t x = 0; x < kPhases; x++) pm_set(n2, x, block_fn(r, pm_at(f, x)));
        f = n2;
    }
    int cur = 0;
    fw[0][t][0] = f.w[0], fw[0][t][1] = f.w[1], fw[0][t][2] = f.w[2];
    for (uint32_t d = 1; d < (uint32_t)kDecBlock; d <<= 1) { // inclusive: thread t's function = chunks 0..t
        __syncthreads();
        PhaseMap a = f;
        if (t >= d) {
            PhaseMap p;
            p.w[0] = fw[cur][t - d][0], p.w[1] = fw[cur][t - d][1], p.w[2] = fw[cur][t - d][2];
            a = pm_compose(p, f);
        }
        f = a;
        cur ^= 1;
        fw[cur][t][0] = f.w[0], fw[cur][t][1] = f.w[1], fw[cur][t][2] = f.w[2];
    }
    __syncthreads();
    // the phase the range is entered in: the file's first workgroup starts on the stream's first token; a later piece of a file
    // where the piece in front ended
    uint32_t ph = ra == fb0 ? recs[fb0].entry_rel : (recs[ra - 1].exit_rel & 31u);
    if (t) {
        PhaseMap p;
        p.w[0] = fw[cur][t - 1][0], p.w[1] = fw[cur][t - 1][1], p.w[2] = fw[cur][t - 1][2];
        ph = pm_at(p, ph);
    }
    for (uint32_t b = i0; b < i1; b++) {
        recs[ra + b].want_rel = ph;
        ph = block_fn(recs[ra + b], ph);
    }
}

// ---- dec_offsets_kernel, one workgroup per file: the stream ends with the FIRST end-of-block symbol of the chain (what lies
//      behind it is padding and the Adler-32, decoded as garbage by their threads); up to there the chain must hold across the
//      workgroups' borders and no subsequence may be invalid; exclusive scan of the workgroups' byte counts; the total must be
//      the filtered image ----
// the stream must end 4 bytes (the Adler-32) before the IDAT does: eob_rel = where the end-of-block symbol of the file's subsequence
// `sub` ends, in bits behind that subsequence's nominal first bit (reference src/fpng.cpp:2331-2340)
__device__ __forceinline__ uint32_t eob_status(const DecJob &job, uint32_t sub, uint32_t eob_rel)
{
    const uint64_t end_bit = job.first_bit + (uint64_t)sub * kSubBits + eob_rel;
    return kDecSawEob | ((((end_bit + 7) >> 3) + 4 != job.z_bytes) ? kDecBadStream : 0u);
}
__global__ __launch_bounds__(kDecBlock) void dec_offsets_kernel(const DecJob *jobs, const DecBlockRec *recs, const uint32_t *bytes, const uint32_t *eob_rel, uint64_t *block_off,
                                                                uint32_t *status, uint32_t *eob_index)
{
    __shared__ uint64_t sums[kDecBlock];
    __shared__ uint32_t red[4];
    const DecJob &job = jobs[blockIdx.x];
    if (job.mode != 0) return;
    const uint32_t t = threadIdx.x, n = job.n_sub, nb = (n + kSubBlock - 1) / kSubBlock, b0 = job.sub_base / kSubBlock;
    const uint32_t per = (nb + kDecBlock - 1) / kDecBlock;
    const uint32_t i0 = min(t * per, nb), i1 = min(i0 + per, nb);
    uint32_t mine = nb;
    for (uint32_t b = i0; b < i1; b++)
        if (recs[b0 + b].first_eob < (uint32_t)kSubBlock) {
            mine = b;
            break;
        }
    const uint32_t last_blk = block_min<kDecBlock / kWave>(mine, red); // nb: the stream never ends
    const uint32_t last_local = last_blk < nb ? recs[b0 + last_blk].first_eob : 0u;
    // workgroups in front of the last one count whole; of the last one, subsequences 0..last_local
    uint64_t local = 0;
    uint32_t bad = 0;
    for (uint32_t b = i0; b < i1 && b <= last_blk; b++) {
        const DecBlockRec r = recs[b0 + b];
        if ((b && r.entry_rel != recs[b0 + b - 1].exit_rel) || !pm_count(rec_map(r))) bad |= kDecNotConverged; // (an empty map: left unsettled by round 0)
        if (b < last_blk) {
            local += r.sum;
            if (r.first_invalid < (uint32_t)kSubBlock) bad |= kDecBadStream;
            if (r.first_overflow < (uint32_t)kSubBlock) bad |= kDecStalled; // (more records than a subsequence has room for: the CPU decoder's file)
        } else {
            if (r.first_invalid <= last_local) bad |= kDecBadStream;
            if (r.first_overflow <= last_local) bad |= kDecStalled;
        }
    }
    uint32_t tail = 0;
    if (last_blk < nb) {
        for (uint32_t k = t; k <= last_local; k += kDecBlock) tail += bytes[job.sub_base + last_blk * kSubBlock + k];
    } else if (t == 0)
        bad |= kDecBadStream;
    const uint32_t tail_sum = block_sum<kDecBlock / kWave>(tail, red);
    const uint32_t any_bad = block_sum<kDecBlock / kWave>(bad & kDecNotConverged, red) ? kDecNotConverged : 0u;
    const uint32_t any_bad2 = block_sum<kDecBlock / kWave>(bad & kDecBadStream, red) ? kDecBadStream : 0u;
    const uint32_t any_bad3 = block_sum<kDecBlock / kWave>(bad & kDecStalled, red) ? kDecStalled : 0u;
    sums[t] = local;
    __syncthreads();
    if (t == 0) {
        uint64_t acc = 0;
        for (int k = 0; k < kDecBlock; k++) {
            const uint64_t v = sums[k];
            sums[k] = acc;
            acc += v;
        }
        uint32_t st = any_bad | any_bad2 | any_bad3;
        if (acc + tail_sum != (uint64_t)(job.bpl + 1) * job.h) st |= kDecBadStream; // too few or too many pixels
        if (last_blk < nb) st |= eob_status(job, last_blk * kSubBlock + last_local, eob_rel[job.sub_base + last_blk * kSubBlock + last_local]);
        if (st) atomicOr(&status[blockIdx.x], st);
        eob_index[blockIdx.x] = last_blk < nb ? last_blk * kSubBlock + last_local : n;
    }
    __syncthreads();
    uint64_t o = sums[t];
    for (uint32_t b = i0; b < i1 && b <= last_blk; b++) {
        block_off[b0 + b] = o;
        o += recs[b0 + b].sum;
    }
}

// ---- the same for a file that arrives in pieces (fpng_amd_decode_host's streamed form): the blocks [blk_a, blk_b) of ONE file, on top
//      of what the pieces in front left in `carry`; `final`: the file's last piece (a stream that has not ended by then never does) ----
__global__ __launch_bounds__(kDecBlock) void dec_offsets_range_kernel(const DecJob *jobs, uint32_t blk_a, uint32_t blk_b, uint32_t final_piece, const DecBlockRec *recs,
                                                                      const uint32_t *bytes, const uint32_t *eob_rel, uint64_t *block_off, uint32_t *status, uint32_t *eob_index,
                                                                      DecCarry *carry)
{
    __shared__ uint64_t sums[kDecBlock];
    __shared__ uint32_t red[4];
    const DecJob &job = jobs[0];
    const DecCarry in = *carry;
    if (in.done) return; // (the stream has ended in an earlier piece: what these blocks hold lies behind it)
    const uint32_t t = threadIdx.x, nb = blk_b - blk_a, b0 = job.sub_base / kSubBlock + blk_a;
    const uint32_t per = (nb + kDecBlock - 1) / kDecBlock;
    const uint32_t i0 = min(t * per, nb), i1 = min(i0 + per, nb);
    uint32_t mine = nb;
    for (uint32_t b = i0; b < i1; b++)
        if (recs[b0 + b].first_eob < (uint32_t)kSubBlock) {
            mine = b;
            break;
        }
    const uint32_t last_blk = block_min<kDecBlock / kWave>(mine, red); // nb: no end in this piece
    const uint32_t last_local = last_blk < nb ? recs[b0 + last_blk].first_eob : 0u;
    uint64_t local = 0;
    uint32_t bad = 0;
    for (uint32_t b = i0; b < i1 && b <= last_blk; b++) {
        const DecBlockRec r = recs[b0 + b];
        if ((blk_a + b && r.entry_rel != recs[b0 + b - 1].exit_rel) || !pm_count(rec_map(r))) bad |= kDecNotConverged;
        if (b < last_blk) {
            local += r.sum;
            if (r.first_invalid < (uint32_t)kSubBlock) bad |= kDecBadStream;
            if (r.first_overflow < (uint32_t)kSubBlock) bad |= kDecStalled;
        } else {
            if (r.first_invalid <= last_local) bad |= kDecBadStream;
            if (r.first_overflow <= last_local) bad |= kDecStalled;
        }
    }
    uint32_t tail = 0;
    if (last_blk < nb) {
        for (uint32_t k = t; k <= last_local; k += kDecBlock) tail += bytes[job.sub_base + (blk_a + last_blk) * kSubBlock + k];
    } else if (t == 0 && final_piece)
        bad |= kDecBadStream;
    const uint32_t tail_sum = block_sum<kDecBlock / kWave>(tail, red);
    const uint32_t any_bad = block_sum<kDecBlock / kWave>(bad & kDecNotConverged, red) ? kDecNotConverged : 0u;
    const uint32_t any_bad2 = block_sum<kDecBlock / kWave>(bad & kDecBadStream, red) ? kDecBadStream : 0u;
    const uint32_t any_bad3 = block_sum<kDecBlock / kWave>(bad & kDecStalled, red) ? kDecStalled : 0u;
    sums[t] = local;
    __syncthreads();
    if (t == 0) {
        uint64_t acc = in.bytes;
        for (int k = 0; k < kDecBlock; k++) {
            const uint64_t v = sums[k];
            sums[k] = acc;
            acc += v;
        }
        uint32_t st = any_bad | any_bad2 | any_bad3;
        DecCarry out = in;
        out.bytes = acc + tail_sum;
        if (last_blk < nb) {
            if (out.bytes != (uint64_t)(job.bpl + 1) * job.h) st |= kDecBadStream; // too few or too many pixels
            st |= eob_status(job, (blk_a + last_blk) * kSubBlock + last_local, eob_rel[job.sub_base + (blk_a + last_blk) * kSubBlock + last_local]);
            eob_index[0] = (blk_a + last_blk) * kSubBlock + last_local;
            out.done = 1;
        } else if (out.bytes > (uint64_t)(job.bpl + 1) * job.h)
            st |= kDecBadStream;
        if (st) atomicOr(&status[0], st);
        *carry = out;
    }
    __syncthreads();
    uint64_t o = sums[t];
    for (uint32_t b = i0; b < i1 && b <= last_blk; b++) {
        block_off[b0 + b] = o;
        o += recs[b0 + b].sum;
    }
}

// ---- dec_subscan_kernel, one workgroup per kDecSubBlock subsequences ----
__global__ __launch_bounds__(kSubBlock) void dec_subscan_kernel(const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t total_subs, DecSubArrays a,
                                                                const uint64_t *block_off, const uint32_t *status, const uint32_t *eob_index)
{
    __shared__ uint32_t wsum[kSubBlock / kWave];
    const uint32_t blk = first_block + blockIdx.x, g0 = blk * kSubBlock;
    if (g0 >= total_subs) return;
    // (everything this thread will want from memory is asked for at once, in front of the questions whose answers decide whether it
    //  is wanted: the kernel is a chain of round trips otherwise -- 14 400 workgroups of it for 8 x 8K)
    const uint32_t t = threadIdx.x, g = g0 + t, lane = t & 63, wv = t >> 6;
    const uint64_t *mine = a.tok + rec_index(g, 0);
    const uint32_t nb_l = a.bytes[g], info_l = a.info[g];
    const uint64_t e0_l = mine[0], e1_l = mine[8], boff_l = block_off[blk]; // (entries 0 and 1)
    uint32_t local0;
    const DecJob &job = job_of_sub(jobs, n_jobs, g0, local0);
    const uint32_t job_index = (uint32_t)(&job - jobs);
    if (status[job_index] & ~kDecSawEob) return;
    const uint32_t last = eob_index[job_index];
    if (local0 > last) return; // behind the end of the stream
    const uint32_t i = local0 + t;
    const bool active = i < job.n_sub && i <= last;
    const uint32_t nb = active ? nb_l : 0u;
    uint32_t incl = nb;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, kWave);
        if ((int)lane >= o) incl += up;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t before = incl - nb;
    for (uint32_t q = 0; q < wv; q++) before += wsum[q];
    if (!active) return;
    a.rel[g] = before;
    const uint32_t *info = a.info + job.sub_base;
    const uint64_t *tok = a.tok;
    const uint32_t sb = job.sub_base;
    const uint32_t nent = info_nrec(info_l);
    const bool need = nent && needs_lastpx(e0_l, e1_l, nent);
    // (the look back reads entries from a subsequence's last one down: the four of a block of the records' layout in one round trip)
    uint64_t lb4[4] = {0, 0, 0, 0};
    uint32_t lb_sub = 0xFFFFFFFFu, lb_at = 0xFFFFFFFFu;
    auto lookback_entry = [&](uint32_t k, uint32_t e) {
        if (k != lb_sub || (e >> 2) != lb_at) {
            lb_sub = k, lb_at = e >> 2;
            const uint64_t *blk4 = tok + rec_index(sb + k, 0) + (size_t)lb_at * 256u;
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) lb4[q] = blk4[q * 8u];
        }
        const uint32_t q = e & 3u;
        return q == 0 ? lb4[0] : (q == 1 ? lb4[1] : (q == 2 ? lb4[2] : lb4[3]));
    };
    const uint32_t lastpx = need ? lookback_lastpx(i, [&](uint32_t k) { return info_nrec(info[k]); }, lookback_entry) : 0u;
    a.lastpx[g] = lastpx;
    // the windows of dec_unfilter_kernel's tiles whose first byte this subsequence writes: their walk over the records starts here --
    // at its first record, or (a subsequence that covers many windows: decode_core.h, resume points) at the entry that reaches the window
    const uint32_t ncb = dec_col_blocks(job.w, job.src_c, job.dst_c), cbw = dec_col_block_bytes(job.src_c, job.dst_c), stride = job.bpl + 1;
    uint32_t *win = job.win;
    if (nb) {
        const uint64_t off = boff_l + before;
        const bool big = nb >= kResumeMinBytes;
        const uint32_t ncap = min(nent, kRecCap);
        ResumeWalk rw = resume_begin(lastpx);
        // (the walk's entries four at a time -- a block of the records' layout, four loads in flight: one at a time is a round trip each)
        uint64_t e4[4] = {0, 0, 0, 0};
        uint32_t e4_at = 0xFFFFFFFFu;
        const uint64_t *mine0 = tok + rec_index(g, 0);
        auto entry_at = [&](uint32_t k) {
            if ((k >> 2) != e4_at) {
                e4_at = k >> 2;
                const uint64_t *blk4 = mine0 + (size_t)e4_at * 256u;
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) e4[q] = blk4[q * 8u];
            }
            const uint32_t q = k & 3u;
            return q == 0 ? e4[0] : (q == 1 ? e4[1] : (q == 2 ? e4[2] : e4[3]));
        };
        for_windows_starting_in(off, nb, cbw, ncb, stride, job.h, [&](uint32_t y, uint32_t cb) {
            uint4 v = make_uint4(i, kNoResume, 0u, 0u);
            if (big) {
                const uint32_t d = (uint32_t)((uint64_t)y * stride + (cb ? 1u + cb * cbw : 0u) - off);
                resume_seek(rw, d, ncap, entry_at);
                v.y = rw.sk, v.z = rw.sc - d, v.w = rw.sth;
            }
            *(uint4 *)(win + ((size_t)y * ncb + cb) * kWinWords) = v;
        });
    }
}

#ifdef FPNG_DEC_TILE_TIMING // diagnostic build (fpng_amd/build.py --variant tile_timing): when does a tile start, have its rows, know its carry, end?
__device__ unsigned long long g_tile_times[8 * 65536];
#define FPNG_TILE_STAMP(k) do { if (threadIdx.x == 0 && item0 + blockIdx.x < 65536) g_tile_times[8 * (item0 + blockIdx.x) + (k)] = wall_clock64(); } while (0) // (by workgroup number)
#else
#define FPNG_TILE_STAMP(k) do { } while (0)
#endif
// ---- the pass that writes.  A tile of dec_unfilter_kernel = kUnfRows rows x one block of columns; every row piece ("window",
//      decode_core.h) is filled in LDS from the token records of the subsequences that cover it.  The WALKS of a tile -- (window,
//      subsequence that reaches into it) pairs, some four hundred -- are numbered through (a prefix sum over the rows' counts) and
//      dealt to the threads one each: no lane idles because its row has seven subsequences where another has nine.  A walk writes
//      its subsequence's bytes with exact stores and only marks the pixels of long matches (decode_core.h: walk_apply, walk_record);
//      when all walks have ended the marked pixels are filled, a lane per pixel (propagate_matches).  A subsequence that straddles
//      two windows is walked for both. ----
constexpr uint32_t kUnfRows = kDecUnfRows;
constexpr uint32_t kTileData = 16;    // where a row's data bytes begin in its LDS row (a dword boundary; the filter byte of the first column block just in front)
constexpr uint32_t kTilePitch = kTileData + 1024 + 8; // bytes of LDS per row: slack of eight bytes on either side of the window (decode_core.h: Out::put64)
constexpr uint32_t kUnfBlock = 512;   // threads of dec_unfilter_kernel: all of them walk and fill matches, kDecBlock of them own a dword column
constexpr uint32_t kRowMaskWords = 8; // a row's bitmap of marked pixels: 256 pixels
constexpr uint32_t kMaxWalksPerRow = 64; // (a subsequence puts out 39 bytes at the very least: a window of 1024 has 28 walks at most)
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t __attribute__((aligned(1))) u32_any_t;
typedef uint64_t __attribute__((aligned(1))) u64_any_t;
typedef __attribute__((address_space(3))) u32_any_t lds_u32_any; // (a dword / two at any byte address: the hardware takes it)
typedef __attribute__((address_space(3))) u64_any_t lds_u64_any;
struct TileOut {
    lds_u8 *win0; // where window byte 0 stands in the tile
    lds_u32 *bm;  // the row's bitmap
    lds_u32 *epx; // the row's entry pixel (a tail's upper half)
    __device__ __forceinline__ void put64(int32_t pos, uint64_t v) { *(lds_u64_any *)(win0 + pos) = v; }
    __device__ __forceinline__ void entry_px(uint32_t th) { *epx = th; }
    __device__ __forceinline__ void mark(uint32_t lo, uint32_t hi)
    {
        while (lo < hi) {
            const uint32_t d = lo >> 5, end = min(hi, (d + 1u) << 5), width = end - lo;
            const uint32_t mask = width == 32u ? 0xFFFFFFFFu : ((1u << width) - 1u) << (lo & 31u);
            __hip_atomic_fetch_or(bm + d, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            lo = end;
        }
    }
};
#ifndef FPNG_DEC_FILL_BATCH
#define FPNG_DEC_FILL_BATCH 4
#endif
// the rows [y0, y0 + nrows) of column block cb of `job` into `tile`; returns the kEmit* flags of this thread's walks.  s_first[r]:
// number of the first walk of row r (s_first[nrows]: all of them), s_i0[r]: the subsequence it walks.
template <int C>
__device__ __forceinline__ uint32_t fill_tile(const DecJob &job, const DecPlaced &pl, uint32_t last, uint32_t y0, uint32_t nrows, uint32_t cb, uint32_t ncb, uint32_t cbw, lds_u8 *tile,
                                              lds_u32 *bm, lds_u32 *epx, uint32_t *s_first, uint32_t *s_i0, uint32_t *s_res, uint32_t item0)
{
    const uint32_t t = threadIdx.x, stride = job.bpl + 1;
    if (t < (uint32_t)kWave) { // the rows' walks: from the subsequence a window begins in to the one the NEXT window (of the stream) begins in
        uint32_t cnt = 0, i0 = 0;
        uint4 ra = make_uint4(0xFFFFFFFFu, kNoResume, 0u, 0u);
        if (t < nrows) {
            const size_t wi = (size_t)(y0 + t) * ncb + cb;
            ra = *(const uint4 *)(job.win + wi * kWinWords); // (all four words: the resume point travels with the subsequence's number, not a round trip behind it)
            const uint32_t a = ra.x;
            uint32_t b = wi + 1 < (size_t)job.h * ncb ? job.win[(wi + 1) * kWinWords] : 0xFFFFFFFFu; // (none: the window nobody begins in -- behind the stream's end, or not placed yet)
            b = min(b, min(last, pl.sub_limit - 1u));
            if (a != 0xFFFFFFFFu && a <= b) cnt = min(b - a + 1u, kMaxWalksPerRow), i0 = a;
        }
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o, kWave);
            if ((int)t >= o) incl += up;
        }
        if (t == 0) s_first[0] = 0;
        s_first[t + 1] = incl, s_i0[t] = i0;
        s_res[t] = ra.y, s_res[kWave + t] = ra.z, s_res[2 * kWave + t] = ra.w;
    }
    for (uint32_t k = t; k < kUnfRows * kRowMaskWords; k += kUnfBlock) bm[k] = 0;
    __syncthreads();
    FPNG_TILE_STAMP(5);
    const uint32_t total = s_first[nrows];
    uint32_t err = 0;
    for (uint32_t q0 = 0; q0 < total; q0 += kUnfBlock) {
        const uint32_t q = q0 + t;
        bool live = q < total;
        uint32_t r = 0;
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1) {
            const uint32_t m = r + step;
            if (m < nrows && s_first[min(m, kUnfRows)] <= q) r = m;
        }
        const uint32_t i = live ? s_i0[r] + (q - s_first[r]) : 0u;
        const Window w = window_of(y0 + r, cb, cbw, stride);
        TileOut out;
        out.win0 = tile + r * kTilePitch + (cb ? kTileData : kTileData - 1u), out.bm = bm + r * kRowMaskWords, out.epx = epx + r;
        const uint32_t g = job.sub_base + i;
        const uint64_t off = pl.block_off[g / kSubBlock] + pl.a.rel[g];
        const uint32_t th0 = pl.a.lastpx[g], info = pl.a.info[g];
        // (a row's first walk: the window's resume point, if its subsequence left one)
        const bool first = live && q == s_first[r];
        const uint4 rs = make_uint4(0u, first ? s_res[r] : kNoResume, s_res[kWave + r], s_res[2 * kWave + r]);
        const bool resumed = first && rs.y != kNoResume;
        live = live && off < w.ws + w.wlen; // (offsets rise: a subsequence that begins behind the window has nothing for it)
        WalkState st;
        st.c = st.c0 = resumed ? (int32_t)rs.z : (live ? (int32_t)(int64_t)(off - w.ws) : (int32_t)w.wlen + 8); // (a lane without a walk: its stores go to the slack behind the row)
        // (a resumed walk's stores may begin in front of the entry it begins with: those bytes -- stale ones of the tail -- lie in front of
        //  the window; held back to that entry's first byte, a store would leave zeros BEHIND a subsequence that ends within eight bytes)
        if (resumed) st.c0 -= 8;
        st.tl = 0, st.th = resumed ? rs.w : th0, st.err = 0;
        const uint32_t nent = live ? min(info_nrec(info), kRecCap) : 0u, kstart = resumed ? rs.y : 0u;
        const gu64e *col = (const gu64e *)(uintptr_t)(pl.a.tok + rec_index(g, 0));
        constexpr uint32_t kBatch = FPNG_DEC_FILL_BATCH; // entries in flight per thread
        static_assert(kResumeAlign % kBatch == 0, "a resumed walk begins with a whole batch");
        uint32_t k = kstart; // (every lane at its own: a resumed walk begins further on)
        // one batch: its entries loaded (all loads in flight, none behind a branch, their addresses one base and constants: what lies
        // behind the subsequence's last entry -- rows that the wave's other lanes mostly need anyway -- is read and counts as nothing;
        // the NEXT batch's loads are issued in front of this one's walk), then walked: all lanes' entries without a match (three steps in four of a
        // gradient, nine in ten of a photograph) -- two groups of literals, one store; one-pixel matches among them -- the same with the
        // tail's pixel for the match; long matches only (flat content) -- their pixels marked; long matches between those (dithered
        // panels) -- marked, the rest one store; anything else (matches of two pixels) -- record by record.  hold: walk_apply (the first batch: a walk's first eight bytes)
        uint64_t rr[kBatch], nx[kBatch]; // the batch at hand, the next one (on its way while this one is walked)
        auto load = [&](uint64_t (&dst)[kBatch], uint32_t kk) {
            const gu64e *ck = col + (size_t)(kk >> 2) * 256u;
#pragma unroll
            for (uint32_t j = 0; j < kBatch; j++) dst[j] = ck[(j >> 2) * 256u + (j & 3u) * 8u];
        };
        auto batch = [&](auto hold, bool act) {
            constexpr bool Hold = decltype(hold)::value;
            const uint32_t left = act ? nent - k : 0u; // entries of this batch that count
#pragma unroll
            for (uint32_t j = 0; j < kBatch; j++) {
                const uint64_t en = j < left ? rr[j] : 0ull;
                const uint32_t a = (uint32_t)en, b = (uint32_t)(en >> 32);
                if (__builtin_amdgcn_ballot_w64(((a | b) & kRecRun) != 0) == 0)
                    walk_entry_literals<Hold>(a, b, st, w, out);
                else if (__builtin_amdgcn_ballot_w64(!entry_plain<C>(a, b)) == 0)
                    walk_entry_plain<C, Hold>(a, b, st, w, stride, out);
                else if (__builtin_amdgcn_ballot_w64(!entry_long_matches(a, b)) == 0)
                    walk_entry_long<C>(a, b, st, w, stride, out);
                else if (__builtin_amdgcn_ballot_w64(!entry_mixed<C>(a, b)) == 0)
                    walk_entry_mixed<C>(a, b, st, w, stride, out);
                else
                    walk_entry<C>(en, st, w, stride, out);
            }
        };
        static_assert(kResumeAlign % kBatch == 0, "a resumed walk begins with a whole batch");
        load(rr, k);
        for (uint32_t it = 0;; it++) {
            const bool act = k < nent && st.c < (int32_t)w.wlen;
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;
            const uint32_t knext = k + (act ? kBatch : 0u); // (a lane that has ended stays where it is)
            load(nx, knext);
            if (it * kBatch < 8u) // (a walk's first eight entries: eight bytes at least)
                batch(std::true_type{}, act);
            else {
                if (!act) st.c = (int32_t)w.wlen + 8; // (a lane that has ended stores into the slack from now on: its subsequence may have been shorter than a store)
                batch(std::false_type{}, act);
            }
            k = knext;
#pragma unroll
            for (uint32_t j = 0; j < kBatch; j++) rr[j] = nx[j];
        }
        err |= st.err;
    }
    FPNG_TILE_STAMP(6);
    return err;
}
// The pixels that the walks marked -- long matches -- take the value of the nearest unmarked pixel to their left in their row (an
// unmarked pixel is a literal one, or a short match's, written by a walk; none in the window: the row's entry pixel).  Items of
// (row, 64 pixels); a wave looks at its items' bitmaps at once (a lane each) and then works through those that have marked pixels.
template <int C> __device__ __forceinline__ void propagate_matches(lds_u8 *tile, const lds_u32 *bm, const lds_u32 *epx, uint32_t nrows, uint32_t *s_front)
{
    const uint32_t lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    constexpr uint32_t kWaves = kUnfBlock / kWave, kItems = kUnfRows * 4, kPerWave = (kItems + kWaves - 1) / kWaves;
    static_assert(kPerWave <= (uint32_t)kWave, "a lane per item");
    // (per row, by a thread each: the nearest unmarked pixel in front of the row's second, third and fourth 64 pixels -- number + 1, 0:
    //  none -- 9 bits each: what an item whose first pixels are marked would otherwise look for through the bitmaps in front of it)
    if (threadIdx.x < nrows) {
        const lds_u32 *m = bm + threadIdx.x * kRowMaskWords;
        uint32_t last = 0, packed = 0;
        for (uint32_t gg = 0; gg < 3; gg++) {
            const uint64_t free = ~((uint64_t)m[gg * 2 + 1] << 32 | m[gg * 2]);
            if (free) last = 64u * gg + 64u - (uint32_t)__builtin_clzll(free);
            packed |= last << (9u * gg);
        }
        s_front[threadIdx.x] = packed;
    }
    __syncthreads();
    uint32_t mlo = 0, mhi = 0;
    {
        const uint32_t it = wv + kWaves * lane, row = it >> 2, grp = it & 3u;
        if (lane < kPerWave && row < nrows) mlo = bm[row * kRowMaskWords + grp * 2], mhi = bm[row * kRowMaskWords + grp * 2 + 1];
    }
    uint64_t todo = __builtin_amdgcn_ballot_w64((mlo | mhi) != 0);
    while (todo) {
        const uint32_t j = (uint32_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        const uint64_t M = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mhi, (int)j) << 32 | (uint32_t)__builtin_amdgcn_readlane((int)mlo, (int)j);
        const uint32_t it = wv + kWaves * j, row = it >> 2, grp = it & 3u;
        const int32_t fsrc = grp ? (int32_t)((s_front[row] >> (9u * (grp - 1u))) & 511u) - 1 : -1; // the nearest unmarked pixel in front of this group
        const uint64_t z = ~M & ((1ull << lane) - 1ull);
        const int32_t src = z ? (int32_t)(64u * grp + 63u - (uint32_t)__builtin_clzll(z)) : fsrc;
        if ((M >> lane) & 1ull) {
            lds_u8 *rowp = tile + row * kTilePitch + kTileData;
            uint32_t v = tail_px<C>(epx[row]);
            if (src >= 0) v = C == 4 ? ((lds_u32 *)rowp)[src] : *(lds_u32_any *)(rowp + 3 * src);
            const uint32_t p = 64u * grp + lane;
            if (C == 4)
                ((lds_u32 *)rowp)[p] = v;
            else
                rowp[3 * p] = (uint8_t)v, rowp[3 * p + 1] = (uint8_t)(v >> 8), rowp[3 * p + 2] = (uint8_t)(v >> 16);
        }
    }
}

// ---- Up filter undone: out[y] = out[y-1] + filtered[y] (bytes, mod 256), every row read ONCE.  One workgroup per kDecBlock dword
//      columns (four byte columns each: packed byte adds) and SEGMENT of kUnfRows rows, which it holds in registers: it adds them
//      up, publishes the segment's column sums, collects the sums of the segments above -- a decoupled look-back: every column sum
//      travels as ONE 8-byte {tag, value} granule written with an agent-scope store (tag = call epoch << 2 | 1: the segment's own
//      sum, | 2: the sum of everything down to its last row), so the value is its own flag and no fence is needed
//      (cdna_hip_programming.md, publish/consume recipe R2); a workgroup only ever waits for workgroups with a LOWER ticket, and
//      tickets are drawn in the order the workgroups start -- then writes the pixels, 3 <-> 4 channels on the way out.  The rows sit
//      in the tile (LDS), filled from the token records (above).  The filter literal in front of every row must be
//      0, then 2 = Up (reference src/fpng.cpp:2255-2259): one lane of the first column block looks. ----
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ uint32_t add_bytes(uint32_t a, uint32_t b)
{
    return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u);
}
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ void store_u32_unaligned(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
// (a wave-uniform base in global memory + a 32-bit offset per lane: the load / store takes the base from scalar registers and ONE
//  vector register of offsets serves all rows -- generic 64-bit addresses cost a register pair per row and lane)
__device__ __forceinline__ uint32_t uni32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uni64(uint64_t x) { return ((uint64_t)uni32((uint32_t)(x >> 32)) << 32) | uni32((uint32_t)x); }
typedef __attribute__((address_space(1))) uint8_t gu8;
typedef uint32_t __attribute__((aligned(1))) u32_any;
typedef __attribute__((address_space(1))) u32_any gu32_any;
// (the empty asm pins the base in scalar registers and hides how it was made: the optimiser otherwise folds "row base + lane
//  offset" into the lane's side of the sum and is back at 64-bit vector addresses)
__device__ __forceinline__ gu8 *scalar_base(const gu8 *p)
{
    uint64_t a = (uint64_t)p;
    asm("" : "+s"(a));
    return (gu8 *)a;
}
__device__ __forceinline__ uint32_t gload_u32(const gu8 *base, uint32_t off) { return *(const gu32_any *)(scalar_base(base) + off); }
__device__ __forceinline__ void gstore_u32(gu8 *base, uint32_t off, uint32_t v) { *(gu32_any *)(scalar_base(base) + off) = v; }
__device__ __forceinline__ void gstore_u8(gu8 *base, uint32_t off, uint32_t v) { scalar_base(base)[off] = (uint8_t)v; }
// (three workgroups of eight waves per compute unit: the tiles' LDS)
__global__ __launch_bounds__(kUnfBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void dec_unfilter_kernel(const DecJob *jobs, DecUnfPlan plan, DecPlaced placed, uint32_t item0, uint32_t *status, uint32_t epoch, uint32_t skip_mask)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile_mem[kUnfRows * kTilePitch];
    __shared__ uint32_t mask_mem[kUnfRows * kRowMaskWords], epx_mem[kUnfRows]; // the rows' marked pixels (long matches), their entry pixels
    __shared__ uint32_t s_first[kWave + 1], s_i0[kWave], s_res[3 * kWave];      // the tile's walks (fill_tile)
    FPNG_TILE_STAMP(0);
    // Items are numbered SEGMENT by segment across all files of the group: the files, sorted by their segment counts (most
    // first), form `pieces` of segments over which the set of files that still have rows is constant -- its first `alive` ones,
    // cbpre[] = their column blocks' prefix sums.  One workgroup per item, item = workgroup number: an item waits for items with
    // LOWER numbers only, and the hardware starts the workgroups of a grid in rising order (per XCD, each XCD taking a fixed share
    // of the numbers: the lowest unfinished item is then always running or next in line for a free slot).  Measured alternatives,
    // 8 x 8K: tickets drawn from one atomic counter 1.72 ms, from one counter per column block 0.78 ms, persistent workgroups
    // taking their items in rising order 0.89 ms, this 0.54 ms, the former two kernels (sums, then a second read) 0.88 ms.  A spin
    // that does not end -- it cannot, unless workgroups do not start in that order after all -- gives up after kSpinLimit polls
    // and leaves the file to the CPU decoder (FPNG_AMD_DECODE_UNDECIDED).
    constexpr uint32_t kSpinLimit = 1u << 20;
    {
        // Which item?  The hardware deals a grid's workgroups to the eight XCDs in turn, and neighbouring items -- the column blocks of
        // one band of rows -- read the same blocks of token records where their windows meet: inside every run of 64 workgroups the
        // numbers are dealt so that eight neighbours share an XCD, i.e. an L2.  (Items still wait for lower numbers only; the order of
        // the runs is the grid's.)
        const uint32_t b = blockIdx.x, b_run = b & ~63u;
        const uint32_t item = item0 + (b_run + 64u <= gridDim.x ? b_run + ((b & 7u) << 3) + ((b >> 3) & 7u) : b); // (item0: a later launch for the same files, fpng_amd_decode_host's streamed form)
        if (item >= plan.total_items) return;
        // (which piece, which file: the lists are short -- eight files, one piece -- and a binary search over them in memory is a chain of
        //  round trips in front of everything else the tile does: the lanes of a wave look at an element each, all at once)
        const uint32_t l64 = threadIdx.x & (kWave - 1);
        DecUnfPiece pc;
        uint32_t per_seg, fidx, cb0, ji_w = 0xFFFFFFFFu;
        if (plan.n_pieces <= (uint32_t)kWave && plan.n_files < (uint32_t)kWave) {
            const bool hasp = l64 < plan.n_pieces, hasf = l64 <= plan.n_files;
            const DecUnfPiece mine = plan.pieces[hasp ? l64 : 0u];
            const uint32_t cbl = plan.cbpre[hasf ? l64 : 0u], ordl = plan.order[l64 < plan.n_files ? l64 : 0u];
            const uint32_t pi = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(hasp && mine.item0 <= item)) - 1u; // (pieces rise; the first one begins at item 0)
            pc.item0 = (uint32_t)__builtin_amdgcn_readlane((int)mine.item0, (int)pi), pc.seg0 = (uint32_t)__builtin_amdgcn_readlane((int)mine.seg0, (int)pi);
            pc.alive = (uint32_t)__builtin_amdgcn_readlane((int)mine.alive, (int)pi), pc.pad_ = 0;
            per_seg = (uint32_t)__builtin_amdgcn_readlane((int)cbl, (int)pc.alive);
            const uint32_t within0 = (item - pc.item0) % per_seg;
            fidx = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(l64 < pc.alive && cbl <= within0)) - 1u;
            cb0 = (uint32_t)__builtin_amdgcn_readlane((int)cbl, (int)fidx);
            ji_w = (uint32_t)__builtin_amdgcn_readlane((int)ordl, (int)fidx);
        } else {
            uint32_t lo = 0, hi = plan.n_pieces;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (plan.pieces[mid].item0 <= item) lo = mid; else hi = mid;
            }
            pc = plan.pieces[lo];
            per_seg = plan.cbpre[pc.alive];
            const uint32_t within0 = (item - pc.item0) % per_seg;
            lo = 0, hi = pc.alive;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (plan.cbpre[mid] <= within0) lo = mid; else hi = mid;
            }
            fidx = lo, cb0 = plan.cbpre[lo];
        }
        const uint32_t rel = item - pc.item0, sg = uni32(pc.seg0 + rel / per_seg), within = rel % per_seg;
        const uint32_t ji = ji_w != 0xFFFFFFFFu ? ji_w : uni32(plan.order[fidx]), cb = uni32(within - cb0);
        // (the file's record, read by every lane, into scalar registers: the compiler keeps what it loads from writable global
        //  memory in vector registers, and every address derived from it would cost a register pair per row)
        DecJob job = jobs[ji];
        job.win = (uint32_t *)uni64((uint64_t)(uintptr_t)job.win), job.out = (uint8_t *)uni64((uint64_t)(uintptr_t)job.out);
        job.segsum = (uint32_t *)uni64((uint64_t)(uintptr_t)job.segsum);
        job.sub_base = uni32(job.sub_base);
        job.w = uni32(job.w), job.h = uni32(job.h), job.bpl = uni32(job.bpl), job.src_c = uni32(job.src_c), job.dst_c = uni32(job.dst_c), job.nseg = uni32(job.nseg), job.mode = uni32(job.mode);
        // (only bits that the kernels in FRONT of this one set decide: every workgroup must come to the same conclusion about a
        //  file, or a later segment would wait for an earlier one that was skipped -- the checks below have bits of their own.
        //  skip_mask = those bits; 0 where kernels that set them run NEXT to this launch -- the streamed form undoes a piece's rows
        //  on a stream of its own while the next piece is decoded: there every workgroup runs and publishes, whatever the status
        //  word says by then; a damaged file's rows are garbage either way and its status says so)
        if (job.mode != 0 || (status[ji] & skip_mask)) return;
        const uint32_t ncol = (job.bpl + 3) / 4;
        const uint32_t sc = job.src_c, dc = job.dst_c, lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
        // Which dword column is this thread's?  Rows of 4-byte pixels: workgroup-thread t has column cb * 256 + t.  Rows of 3-byte
        // pixels: a wave takes 48 dword columns = 192 bytes = 64 WHOLE pixels (its lanes 48..63 hold no column), a column block is
        // 256 pixels either way; where such rows become 4-channel pixels every lane writes one pixel, gathered from two lanes'
        // dwords -- dword stores instead of a byte at a time.
        const bool three = sc == 3, widen = three && dc == 4;
        const uint32_t wave_px = (cb * (kDecBlock / kWave) + wv) * kWave; // (the wave's first pixel)
        const uint32_t j4 = three ? (wave_px / 4) * 3 + lane : cb * kDecBlock + threadIdx.x;
        const bool active = j4 < ncol && (!three || lane < 48) && threadIdx.x < (uint32_t)kDecBlock;
        const uint32_t y0 = sg * kUnfRows, nrows = min(kUnfRows, job.h - y0);
        // ---- the tile's rows, from the token records (all threads; the barriers stand in front of every way out) ----
        lds_u8 *tile = (lds_u8 *)tile_mem;
        {
            const uint32_t ncb = dec_col_blocks(job.w, sc, dc), cbw = dec_col_block_bytes(sc, dc), last_sub = placed.eob_index[ji];
            lds_u32 *bm = (lds_u32 *)mask_mem, *epx = (lds_u32 *)epx_mem;
            const uint32_t err = sc == 4 ? fill_tile<4>(job, placed, last_sub, y0, nrows, cb, ncb, cbw, tile, bm, epx, s_first, s_i0, s_res, item0)
                                         : fill_tile<3>(job, placed, last_sub, y0, nrows, cb, ncb, cbw, tile, bm, epx, s_first, s_i0, s_res, item0);
            if (err) atomicOr(&status[ji], err);
            __syncthreads();
            if (sc == 4) propagate_matches<4>(tile, bm, epx, nrows, s_i0); else propagate_matches<3>(tile, bm, epx, nrows, s_i0);
        }
        __syncthreads();
        FPNG_TILE_STAMP(1);
        if (threadIdx.x >= (uint32_t)kDecBlock) return; // (the tile is filled: from here on a thread per dword column)
        if (cb == 0 && threadIdx.x == 0) {
            bool bad = false;
            for (uint32_t k = 0; k < nrows; k++) bad |= tile[k * kTilePitch + kTileData - 1] != (y0 + k ? 2 : 0);
            if (bad) atomicOr(&status[ji], kDecBadFilter);
        }
        if (!widen && !active) return; // (the lanes of a widening wave all stay: they write pixels)
        if (widen && wave_px >= job.w) return;
        // ---- the columns' running sums, in place: row k of the tile becomes the sum of its rows 0 .. k (every thread its own dword
        //      column; eight rows in flight).  The rows stay in LDS -- until round 6 a thread held its 48 of them in registers, which
        //      is what kept the kernel at four waves per SIMD. ----
        lds_u32 *T = (lds_u32 *)(tile + kTileData) + (three ? wv * 48 + lane : threadIdx.x); // this thread's dword column of the tile
        constexpr uint32_t P = kTilePitch / 4;
        uint32_t p = 0;
        if (active) {
            for (uint32_t k = 0; k < nrows; k += 8) {
                uint32_t t8[8];
#pragma unroll
                for (uint32_t q = 0; q < 8; q++) t8[q] = T[min(k + q, nrows - 1) * P];
#pragma unroll
                for (uint32_t q = 0; q < 8; q++)
                    if (k + q < nrows) p = add_bytes(p, t8[q]), T[(k + q) * P] = p;
            }
        }
        FPNG_TILE_STAMP(2);
        gu64 *gran = (gu64 *)(uintptr_t)job.segsum + j4;
        uint32_t carry = 0;
        if (active && (sg + 1 < job.nseg || sg)) { // (a file of one segment publishes nothing)
            gu64 *mine = gran + (size_t)sg * ncol;
            if (sg == 0)
                __hip_atomic_store(mine, ((unsigned long long)(epoch << 2 | 2u) << 32) | p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else {
                if (sg + 1 < job.nseg) __hip_atomic_store(mine, ((unsigned long long)(epoch << 2 | 1u) << 32) | p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool stalled = false;
                for (uint32_t q = sg; q-- > 0 && !stalled;) {
                    gu64 *g = gran + (size_t)q * ncol;
                    unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (uint32_t spins = 0; (uint32_t)(x >> 34) != epoch; spins++) {
                        if (spins == kSpinLimit) {
                            stalled = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    carry = add_bytes(carry, (uint32_t)x);
                    if (((uint32_t)(x >> 32) & 3u) == 2u) break;
                }
                if (stalled) atomicOr(&status[ji], kDecStalled);
                if (sg + 1 < job.nseg)
                    __hip_atomic_store(mine, ((unsigned long long)(epoch << 2 | 2u) << 32) | add_bytes(carry, p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- the pixels.  Dword stores at any byte address (rows of 3-channel pixels start anywhere; the hardware takes
        //      unaligned dwords, as it does for the loads above); bytes only where a row ends inside a dword ----
        FPNG_TILE_STAMP(3);
        const size_t os = (size_t)job.w * dc;
        gu8 *orow = (gu8 *)(uintptr_t)(job.out + (size_t)y0 * os);
        auto row_sum = [&](uint32_t k) { return active ? add_bytes(carry, T[k * P]) : 0u; }; // the pixels' bytes of row k, this thread's four
        if (sc == dc) {
            const uint32_t nb = min(4u, job.bpl - j4 * 4);
            if (nb == 4) {
                for (uint32_t k = 0; k < nrows; k++) gstore_u32(orow + (size_t)k * os, j4 * 4, row_sum(k));
            } else {
                for (uint32_t k = 0; k < nrows; k++) {
                    const uint32_t v = row_sum(k);
                    for (uint32_t b = 0; b < nb; b++) gstore_u8(orow + (size_t)k * os, j4 * 4 + b, v >> (8 * b));
                }
            }
        } else if (widen) {
            // lane L's pixel = bytes 3L .. 3L + 2 of the wave's 192: in the dwords of lanes 3L / 4 and the next one
            const uint32_t src = (3u * lane) >> 2, sh = (3u * lane) & 3u;
            const bool st = wave_px + lane < job.w;
            for (uint32_t k = 0; k < nrows; k++) {
                const uint32_t v = row_sum(k);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v), hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src + 1) << 2), (int)v);
                if (st) gstore_u32(orow + (size_t)k * os, (wave_px + lane) * 4, funnel(hi, lo, 8 * sh) | 0xFF000000u);
            }
        } else {
            // 4 -> 3 channels: the wave's 64 pixels are 48 dwords; lane L < 48 builds dword L = bytes 4L .. 4L + 3 of the 192
            // from the pixels 4L / 3 and the next one
            const uint32_t wpx = (cb * (kDecBlock / kWave) + wv) * kWave, nv = min((uint32_t)kWave, job.w - wpx); // (this wave's pixels: j4 = wpx + lane < w)
            const uint32_t p0 = (4u * lane) / 3u, r = 4u * lane - 3u * p0, have = 3u * nv; // bytes of the wave
            const uint32_t nb = lane < 48 ? (4u * lane + 4 <= have ? 4u : (4u * lane < have ? have - 4u * lane : 0u)) : 0u;
            const uint32_t off = wpx * 3 + 4u * lane;
            auto dword = [&](uint32_t acc) {
                const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(p0 << 2), (int)acc), b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((p0 + 1) & 63u) << 2), (int)acc);
                return funnel((b & 0xFFFFFFu) >> 8, (a & 0xFFFFFFu) | (b << 24), 8 * r); // (b's 24 bits : a's 24 bits) >> 8 r
            };
            const bool ragged = __builtin_amdgcn_ballot_w64(nb - 1u < 3u) != 0; // the row ends inside some lane's dword (all lanes come along: they are the gather's sources)
            for (uint32_t k = 0; k < nrows; k++) {
                const uint32_t d = dword(row_sum(k));
                if (nb == 4) gstore_u32(orow + (size_t)k * os, off, d);
                if (ragged && nb - 1u < 3u)
                    for (uint32_t q = 0; q < nb; q++) gstore_u8(orow + (size_t)k * os, off + q, d >> (8 * q));
            }
        }
        FPNG_TILE_STAMP(4);
    }
}

// ---- stored files (reference src/fpng.cpp:2107-2207): the filter-0 stream sits in stored blocks, in fpng's files all of 65535
//      bytes but the last.  The kernel copies the rows out of that layout -- a workgroup takes rows in turn, its threads a dword
//      (or, where the channel count changes, a pixel) each -- and CHECKS the layout while it is there: every block header at the
//      place the usual layout has it, every row's filter byte 0.  Anything else sets kDecStoredOdd, and the host looks at the file
//      itself (a valid file with other block sizes is the CPU decoder's; check_stored() in decode_api.cpp is the rule). ----
__device__ __forceinline__ uint64_t stored_pos(uint64_t s) // byte s of the stream -> its offset in the zlib data
{
    uint64_t q = s >> 16; // s / 65535, from below: s = 65536 q + lo = 65535 q + (q + lo)
    uint64_t r = q + (s & 0xFFFFu);
    while (r >= 65535) r -= 65535, q++;
    return 2 + 5 * (q + 1) + s;
}
__global__ __launch_bounds__(kDecBlock) void dec_stored_kernel(const DecJob *jobs, uint32_t *status)
{
    const DecJob &job = jobs[blockIdx.y];
    if (job.mode != 1) return;
    const uint8_t *z = job.z + job.z_shift;
    const uint32_t sc = job.src_c, dc = job.dst_c, bpl = job.bpl, w = job.w, h = job.h;
    const uint64_t total = ((uint64_t)bpl + 1) * h;
    const uint32_t nblk = (uint32_t)((total + 65534) / 65535);
    bool odd = false;
    // block headers: final flag | type 0, length, its complement
    for (uint32_t i = blockIdx.x * kDecBlock + threadIdx.x; i < nblk; i += gridDim.x * kDecBlock) {
        const uint8_t *hd = z + 2 + (uint64_t)i * 65540;
        const uint32_t len = i + 1 < nblk ? 65535u : (uint32_t)(total - (uint64_t)i * 65535);
        odd |= hd[0] != (i + 1 == nblk ? 1 : 0) || (hd[1] | hd[2] << 8) != len || (hd[3] | hd[4] << 8) != (~len & 0xFFFFu);
    }
    // rows: `lanes` threads per row (a power of two), kDecBlock / lanes rows per workgroup and turn
    const uint32_t units = sc == dc ? (bpl + 3) / 4 : w; // dwords of a row, or pixels
    uint32_t lanes = 1;
    while (lanes < units && lanes < (uint32_t)kDecBlock) lanes <<= 1;
    const uint32_t rows_per = kDecBlock / lanes, t_row = threadIdx.x / lanes, t_x = threadIdx.x & (lanes - 1);
    const size_t os = (size_t)w * dc;
    for (uint32_t y = blockIdx.x * rows_per + t_row; y < h; y += gridDim.x * rows_per) {
        const uint64_t s0 = (uint64_t)y * ((uint64_t)bpl + 1); // the row's filter byte
        if (t_x == 0) odd |= z[stored_pos(s0)] != 0;
        uint8_t *o = job.out + (size_t)y * os;
        if (sc == dc) {
            for (uint32_t x = 4 * t_x; x < bpl; x += 4 * lanes) {
                const uint64_t s = s0 + 1 + x, p = stored_pos(s);
                const uint32_t nb = min(4u, bpl - x);
                if (nb == 4 && stored_pos(s + 3) == p + 3)
                    store_u32_unaligned(o + x, load_u32_unaligned(z + p));
                else
                    for (uint32_t b = 0; b < nb; b++) o[x + b] = z[stored_pos(s + b)];
            }
        } else {
            for (uint32_t x = t_x; x < w; x += lanes) {
                const uint64_t s = s0 + 1 + (uint64_t)x * sc;
                uint32_t px = 0xFF000000u;
                for (uint32_t b = 0; b < 3; b++) px |= (uint32_t)z[stored_pos(s + b)] << (8 * b);
                if (dc == 4)
                    store_u32_unaligned(o + (size_t)x * 4, px);
                else
                    for (uint32_t b = 0; b < 3; b++) o[(size_t)x * 3 + b] = (uint8_t)(px >> (8 * b));
            }
        }
    }
    if (odd) atomicOr(&status[blockIdx.y], kDecStoredOdd);
}

// ---- the kernels' lookup table (decode_core.h) from a file's 288 literal / length code lengths, one workgroup per table: what
//      build_multi_lut() (decode_api.cpp) does on the host -- a batch of 2-pass files has a table per file, and building and
//      uploading 16 KB for each kept the host busy longer than the GPU decoded.  The lengths were checked by the host (a complete
//      code of at most 12-bit codes, or a single code: png_parse.h).  Canonical codes (RFC 1951 3.2.2): symbols sorted by (length,
//      value); entry k of the table is found by trying the 12 lengths on k's low bits. ----
// length symbols 257 .. 285: base length and extra bits (RFC 1951 3.2.5)
__constant__ uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__global__ __launch_bounds__(kDecBlock) void dec_build_lut_kernel(const uint8_t *sizes_all, uint32_t *luts)
{
    __shared__ uint8_t len[288];
    __shared__ uint16_t sorted[288];
    __shared__ uint32_t first[16], count[16], offset[16];
    __shared__ uint32_t t1[kLutEntries]; // symbol | length << 9 per 12-bit index (0: no code)
    const uint8_t *sizes = sizes_all + (size_t)blockIdx.x * 288;
    uint32_t *lut = luts + (size_t)blockIdx.x * kLutDwords;
    const uint32_t t = threadIdx.x;
    for (uint32_t s = t; s < 288; s += kDecBlock) len[s] = sizes[s];
    __syncthreads();
    if (t < 16) {
        uint32_t c = 0;
        for (uint32_t s = 0; s < 288; s++) c += len[s] == t;
        count[t] = t ? c : 0u;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t code = 0, off = 0;
        first[0] = offset[0] = 0;
        for (uint32_t l = 1; l < 16; l++) {
            code = (code + count[l - 1]) << 1;
            first[l] = code, offset[l] = off;
            off += count[l];
        }
    }
    __syncthreads();
    for (uint32_t s = t; s < 288; s += kDecBlock) {
        const uint32_t l = len[s];
        if (!l) continue;
        uint32_t rank = 0;
        for (uint32_t q = 0; q < s; q++) rank += len[q] == l;
        sorted[offset[l] + rank] = (uint16_t)s;
    }
    __syncthreads();
    for (uint32_t k = t; k < kLutEntries; k += kDecBlock) {
        uint32_t e = 0;
        for (uint32_t l = 1; l <= 12; l++) {
            const uint32_t c = __brev(k << (32 - l)); // the low l bits of k, first bit highest: a code of l bits
            if (c >= first[l] && c - first[l] < count[l]) e = sorted[offset[l] + (c - first[l])] | l << 9;
        }
        t1[k] = e;
    }
    __syncthreads();
    for (uint32_t k = t; k < kLutEntries; k += kDecBlock) {
        const uint32_t e1 = t1[k], l1 = (e1 >> 9) & 15u, s1 = e1 & 511u;
        uint32_t ent = 0;
        if (!l1 || s1 > 285)
            ent = 0;
        else if (s1 == 256)
            ent = kEntEob | l1 << 12;
        else if (s1 > 256)
            ent = kLenExtra[s1 - 257] ? kEntMatch | l1 << 12 | (uint32_t)kLenExtra[s1 - 257] << 9 | kLenBase[s1 - 257] : (l1 + 1) << 28 | kEntMatch | kLenBase[s1 - 257];
        else {
            uint32_t L = l1, n = 1, lits = s1;
            while (n < 3) { // the next code is whole if its length fits into the index bits that are left
                const uint32_t e2 = t1[k >> L], l2 = (e2 >> 9) & 15u, s2 = e2 & 511u;
                if (!l2 || s2 >= 256 || L + l2 > 12) break;
                lits |= s2 << (8 * n);
                n++, L += l2;
            }
            ent = L << 28 | n << 26 | lits;
        }
        lut[k] = ent;
    }
    if (t < 64) lut[kLutEntries + t] = len[4 * t] | len[4 * t + 1] << 8 | len[4 * t + 2] << 16 | (uint32_t)len[4 * t + 3] << 24;
}

// ---- device-resident files: the first `head` and the last `tail` bytes of every file gathered into one buffer (one copy to the
//      host instead of two per file) ----
__global__ __launch_bounds__(kDecBlock) void dec_fetch_kernel(const DecFileRef *files, uint32_t head, uint32_t tail, uint8_t *out)
{
    const DecFileRef f = files[blockIdx.x];
    uint8_t *o = out + (size_t)blockIdx.x * (head + tail);
    if (!f.data || !f.size) return;
    const uint32_t hl = min(f.size, head), tl = f.size > hl ? min(f.size - hl, tail) : 0u;
    for (uint32_t k = threadIdx.x; k < hl; k += kDecBlock) o[k] = f.data[k];
    for (uint32_t k = threadIdx.x; k < tl; k += kDecBlock) o[head + k] = f.data[f.size - tl + k];
}

} // namespace

void launch_dec_build_luts(hipStream_t s, const uint8_t *sizes, uint32_t n, uint32_t *luts)
{
    if (n) hipLaunchKernelGGL(dec_build_lut_kernel, dim3(n), dim3(kDecBlock), 0, s, sizes, luts);
}
void launch_dec_fetch(hipStream_t s, const DecFileRef *files, uint32_t n, uint32_t head, uint32_t tail, uint8_t *out)
{
    hipLaunchKernelGGL(dec_fetch_kernel, dim3(n), dim3(kDecBlock), 0, s, files, head, tail, out);
}

// (the synchronisation and the emit run as persistent workgroups, `resident` of them: a few per compute unit)
void launch_dec_sync(hipStream_t s, uint32_t resident, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, uint32_t round,
                     DecSubArrays a, DecBlockRec *recs, uint32_t *changed, uint32_t *multi)
{
    // (the workgroups' maps have more than one pair only behind round 1: no chain in front of it)
    if (round >= 2) hipLaunchKernelGGL(dec_chain_kernel, dim3(n_jobs), dim3(kDecBlock), 0, s, jobs, first_block, n_blocks, recs, multi);
    // a border round has something to do in a few workgroups' blocks only: `resident` workgroups walk over the records (5 us instead of
    // 20 for the 14 400 blocks of 8 x 8K that a workgroup each would look at)
    const dim3 block(kSubBlock);
    if (round)
        hipLaunchKernelGGL(dec_sync_kernel<true>, dim3(std::min(n_blocks, resident)), block, 0, s, jobs, n_jobs, first_block, n_blocks, total_subs, round, a, recs, changed, multi);
    else
        hipLaunchKernelGGL(dec_sync_kernel<false>, dim3(FPNG_DEC_PERSISTENT ? std::min(n_blocks, resident) : n_blocks), block, 0, s, jobs, n_jobs, first_block, n_blocks, total_subs, round, a,
                           recs, changed, multi);
}
void launch_dec_offsets(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, const DecJob *group_jobs,
                        uint32_t n_group_jobs, DecSubArrays a, const DecBlockRec *recs, uint64_t *block_off, uint32_t *status, uint32_t *eob_index)
{
    const uint32_t j0 = (uint32_t)(group_jobs - jobs);
    hipLaunchKernelGGL(dec_offsets_kernel, dim3(n_group_jobs), dim3(kDecBlock), 0, s, group_jobs, recs, a.bytes, a.eob, block_off, status + j0, eob_index + j0);
    hipLaunchKernelGGL(dec_subscan_kernel, dim3(n_blocks), dim3(kSubBlock), 0, s, jobs, n_jobs, first_block, total_subs, a, block_off, status, eob_index);
}
void launch_dec_offsets_range(hipStream_t s, const DecJob *jobs, uint32_t sub_base_block, uint32_t blk_a, uint32_t blk_b, bool final_piece, uint32_t total_subs, DecSubArrays a,
                              const DecBlockRec *recs, uint64_t *block_off, uint32_t *status, uint32_t *eob_index, DecCarry *carry)
{
    hipLaunchKernelGGL(dec_offsets_range_kernel, dim3(1), dim3(kDecBlock), 0, s, jobs, blk_a, blk_b, final_piece ? 1u : 0u, recs, a.bytes, a.eob, block_off, status, eob_index, carry);
    hipLaunchKernelGGL(dec_subscan_kernel, dim3(blk_b - blk_a), dim3(kSubBlock), 0, s, jobs, 1u, sub_base_block + blk_a, total_subs, a, block_off, status, eob_index);
}
// jobs / status: of the group's first file; plan: device arrays (decode_api.cpp); epoch: this launch's (a new one every time; the
// granules are never cleared)
void launch_dec_unfilter(hipStream_t s, const DecJob *jobs, DecUnfPlan plan, DecPlaced placed, uint32_t item0, uint32_t n_items, uint32_t *status, uint32_t epoch, bool concurrent_status)
{
    if (n_items)
        hipLaunchKernelGGL(dec_unfilter_kernel, dim3(n_items), dim3(kUnfBlock), 0, s, jobs, plan, placed, item0, status, epoch,
                           concurrent_status ? 0u : (kDecNotConverged | kDecBadStream | kDecStalled));
}
#ifdef FPNG_DEC_SYNC_TIMING
void dec_dump_sync_times(const char *path, uint32_t n_blocks)
{
    std::vector<unsigned long long> t(8 * (size_t)std::min(n_blocks, 65536u));
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_sync_times), t.size() * 8) != hipSuccess) return;
    if (FILE *f = fopen(path, "w")) {
        for (size_t i = 0; i < t.size() / 8; i++) fprintf(f, "%zu %llu %llu %llu %llu %llu %llu\n", i, t[8 * i], t[8 * i + 1], t[8 * i + 2], t[8 * i + 3], t[8 * i + 4], t[8 * i + 5]);
        fclose(f);
    }
}
#endif
#ifdef FPNG_DEC_TILE_TIMING
void dec_dump_tile_times(const char *path, uint32_t n_items)
{
    std::vector<unsigned long long> t(8 * (size_t)std::min(n_items, 65536u));
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_tile_times), t.size() * 8) != hipSuccess) return;
    if (FILE *f = fopen(path, "w")) {
        for (size_t i = 0; i < t.size() / 8; i++) fprintf(f, "%zu %llu %llu %llu %llu %llu %llu %llu %llu\n", i, t[8 * i], t[8 * i + 1], t[8 * i + 2], t[8 * i + 3], t[8 * i + 4], t[8 * i + 5], t[8 * i + 6], t[8 * i + 7]);
        fclose(f);
    }
}
#endif
void launch_dec_finish(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, DecUnfPlan plan, DecPlaced placed, uint32_t *status, uint32_t epoch, bool any_stored)
{
    if (plan.total_items) launch_dec_unfilter(s, jobs, plan, placed, 0, plan.total_items, status, epoch, false);
    if (!any_stored) return; // (a workgroup that finds its file is not a stored one leaves at once, but n_jobs x 512 of them is not free)
    for (uint32_t j0 = 0; j0 < n_jobs; j0 += 32768) // (the y dimension of a grid holds at most 65535 workgroups)
        hipLaunchKernelGGL(dec_stored_kernel, dim3(512, std::min(32768u, n_jobs - j0)), dim3(kDecBlock), 0, s, jobs + j0, status + j0);
}

} // namespace fpng_amd
