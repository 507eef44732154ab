// decode.hip -- GPU batch decoder of fpng-written PNG files (reference src/fpng.cpp:2209-2901 decodes the same
// streams serially) in its data-parallel form: the container and the Deflate block header are parsed on the host
// (a few hundred bytes per file, decode_api.cpp); everything that touches the pixel stream runs here.
//
// An fpng stream is ONE Huffman-coded bit string with no restart points, but Huffman decoders SELF-SYNCHRONISE: started at a
// wrong bit, a decoder falls into step with the true token sequence after a few dozen bits.  So the token bits are cut into
// subsequences of kSubBits bits, one thread each, kDecSubBlock of them per workgroup (whose slice of the bits and the lookup
// table are staged in LDS):
//   dec_sync_kernel    round 0: every thread decodes its subsequence from its nominal first bit and notes where it crossed
//                      into the next one; rounds 1..R: every thread restarts where its predecessor ended if that differs from
//                      where it started before (in place).  After a round without changes the ends form the TRUE chain from the
//                      stream's first token.  The host launches the first rounds blind; dec_offsets_kernel's chain check says
//                      whether a file needs more (decode_api.cpp gives up on the GPU path for it beyond kMaxRounds).
//   dec_blocksum_kernel / dec_offsets_kernel   per workgroup, then per file: output bytes in front of every workgroup, the
//                      stream's end (first end-of-block symbol of the chain), chain and total checked
//   dec_emit_kernel    decodes again, now for real: literals go to the filtered image; a match (always "repeat the previous
//                      pixel", reference fpng.cpp:2273-2330) only marks its pixels in a bit mask; every rule of the reference's
//                      decoder is checked (filter literal 0 then 2, matches whole pixels inside a row, exact total, EOB, the
//                      stream ends 4 bytes before the IDAT does)
//   dec_fill_kernel    one wave per row: marked pixels take the value of the nearest unmarked pixel to their left
//   dec_unfilter_sums_kernel / dec_unfilter_kernel   the Up filter undone: one thread per dword column and segment of 128 rows
//                      (segment sums first), channel count conversion
//   dec_stored_kernel  files that are stored blocks (reference fpng.cpp:2107-2207): a strided copy
#include "decode.h"

#include <hip/hip_runtime.h>

namespace fpng_amd {

namespace {

constexpr int kWave = 64;
constexpr int kDecBlock = 256;
#define FPNG_DEC_GLOBAL __attribute__((address_space(1)))
constexpr int kSubBlock = (int)kDecSubBlock; // subsequences (= threads) per workgroup of the decoding kernels
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t dec_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The token bits a workgroup works on are staged in LDS first (coalesced loads): thread t of a block decodes the kSubBits bits that
// start kSubBits / 32 dwords behind thread t-1's, so straight from global memory every load instruction of a wave would touch 64
// different cache lines.  A block's slice = its kSubBlock subsequences plus the few dwords the last thread may run over.  In LDS
// every 32 dwords of the slice are followed by a COPY of the next dword: dword d sits at d + d / 32 and its successor always in
// the next slot (one ds_read2 per token), and the lanes of a wave, 16 dwords apart, meet in different banks.
constexpr uint32_t kSliceDwords = kSubBlock * (kSubBits / 32) + 8;
constexpr uint32_t kSliceSlots = kSliceDwords + kSliceDwords / 32 + 2;
__device__ __forceinline__ uint32_t slice_slot(uint32_t d) { return d + (d >> 5); }

// LSB-first reader over the staged slice; positions are bits relative to the slice's first dword.  A token has at most 12 + 5 + 1
// bits: one 32-bit window per token.
struct LdsBits {
    const uint32_t *l;
    uint32_t pos;
    uint32_t limit; // no token may start at or behind this bit
    __device__ __forceinline__ uint32_t window() const
    {
        const uint32_t s = slice_slot(pos >> 5);
        return __builtin_amdgcn_alignbit(l[s + 1], l[s], pos & 31u);
    }
};

enum : uint32_t { kSubEob = 1u, kSubInvalid = 4u };

// one token: returns its kind and advances.  kind: 0..255 literal, 256 end of block, 257.. match with `run` bytes; -1 invalid.
// lut[next 12 bits] = symbol | code length << 9 | (length symbols) extra bits << 13 | base length << 16, 0 = no such code
__device__ __forceinline__ int next_token(LdsBits &in, const uint32_t *lut, uint32_t &run)
{
    const uint32_t w = in.window();
    const uint32_t e = lut[w & 4095u];
    const uint32_t len = (e >> 9) & 15u;
    if (!len) return -1;
    const uint32_t sym = e & 511u;
    if (sym <= 256) {
        in.pos += len;
        return (int)sym;
    }
    const uint32_t xb = (e >> 13) & 7u;
    run = (e >> 16) + ((w >> len) & ((1u << xb) - 1u));
    in.pos += len + xb + 1; // extra bits + the 1-bit distance code ("previous pixel")
    return (int)sym;
}

__device__ __forceinline__ const DecJob &job_of_sub(const DecJob *jobs, uint32_t n_jobs, uint32_t g, uint32_t &local)
{
    // binary search over sub_base (jobs are few; subsequences many)
    uint32_t lo = 0, hi = n_jobs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs[mid].sub_base <= g) lo = mid; else hi = mid;
    }
    local = g - jobs[lo].sub_base;
    return jobs[lo];
}

// lookup table + the block's slice of the token bits into LDS; returns the slice's first bit (absolute, from z)
__device__ __forceinline__ uint64_t stage_block(const DecJob &job, uint32_t local0, uint32_t *lut, uint32_t *bits)
{
    const u32x4 *src = (const u32x4 *)job.lut;
    for (int i = threadIdx.x; i < 1024; i += kSubBlock) ((u32x4 *)lut)[i] = src[i];
    const uint64_t d0 = (job.first_bit + (uint64_t)local0 * kSubBits) >> 5;
    const uint64_t n_dw = (job.z_bytes + 16) >> 2; // (the stream's buffer has 16 spare bytes behind the data)
    const uint32_t *w = (const uint32_t *)job.z_aligned;
    for (uint32_t d = threadIdx.x; d < kSliceDwords; d += kSubBlock) {
        const uint32_t v = (d0 + d < n_dw) ? w[d0 + d] : 0u;
        const uint32_t sl = slice_slot(d);
        bits[sl] = v;
        if (d && !(d & 31u)) bits[sl - 1] = v; // the copy behind the 32 dwords in front
    }
    return d0 << 5;
}

// ---- synchronisation rounds (results updated in place: a thread that reads its predecessor's end while that one is being
//      rewritten decodes again in the next round -- the rounds end when one of them changes nothing) ----
__global__ __launch_bounds__(kSubBlock) void dec_sync_kernel(const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t total_subs, uint32_t round,
                                                             uint64_t *start, uint64_t *end, uint32_t *bytes, uint32_t *flags, uint32_t *changed)
{
    __shared__ __attribute__((aligned(16))) uint32_t lut[4096];
    __shared__ uint32_t bits[kSliceSlots];
    const uint32_t blk = first_block + blockIdx.x, g0 = blk * kSubBlock;
    if (g0 >= total_subs) return;
    // all subsequences of a block belong to one job (sub_base is padded to kSubBlock by the host)
    uint32_t local0;
    const DecJob &job = job_of_sub(jobs, n_jobs, g0, local0);
    const uint32_t g = g0 + threadIdx.x, i = local0 + threadIdx.x;
    const bool valid = i < job.n_sub;
    const uint64_t nominal = job.first_bit + (uint64_t)i * kSubBits;
    uint64_t s = nominal;
    bool work = valid;
    if (round && valid) {
        s = i ? end[g - 1] : job.first_bit;
        work = s != start[g];
    }
    if (!__syncthreads_or(work)) return; // nothing new for this block: no staging either
    const uint64_t base = stage_block(job, local0, lut, bits);
    __syncthreads();
    if (!work) return;
    if (round) atomicOr(changed, 1u);
    start[g] = s;
    LdsBits in;
    in.l = bits;
    const uint64_t lim = job.end_limit_bit - base;
    in.limit = lim > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)lim;
    const uint32_t boundary = (uint32_t)(nominal + kSubBits - base);
    in.pos = (uint32_t)(s - base);
    uint32_t nbytes = 0, fl = 0;
    while (in.pos < boundary) {
        if (in.pos >= in.limit) { // ran off the data without an end-of-block symbol
            fl = kSubInvalid;
            break;
        }
        uint32_t run = 0;
        const int t = next_token(in, lut, run);
        if (t < 0) {
            fl = kSubInvalid;
            break;
        }
        if (t == 256) {
            fl = kSubEob;
            break;
        }
        nbytes += t < 256 ? 1u : run;
    }
    // A decode that derailed or met an end-of-block symbol hands over on the nominal boundary: speculative decodes meet FALSE
    // end-of-block symbols, and letting those stop their successors would cost one round per subsequence to undo.  Which
    // end-of-block symbol is the true one is settled afterwards: the first one of the converged chain (dec_offsets_kernel).
    end[g] = base + (fl ? boundary : in.pos);
    bytes[g] = nbytes;
    flags[g] = fl;
}

// ---- output offsets, in two steps.  dec_blocksum_kernel, one workgroup per 256 subsequences: their output bytes, the first
//      one that met an end-of-block symbol, the first one that does not start where its predecessor ended, the first invalid one ----
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
__global__ __launch_bounds__(kSubBlock) void dec_blocksum_kernel(const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t total_subs, const uint64_t *start,
                                                                 const uint64_t *end, const uint32_t *bytes, const uint32_t *flags, DecBlockRec *recs)
{
    __shared__ uint32_t red[kSubBlock / kWave];
    const uint32_t blk = first_block + blockIdx.x, g0 = blk * kSubBlock;
    if (g0 >= total_subs) return;
    uint32_t local0;
    const DecJob &job = job_of_sub(jobs, n_jobs, g0, local0);
    const uint32_t t = threadIdx.x, g = g0 + t, i = local0 + t;
    const bool valid = i < job.n_sub;
    const uint32_t f = valid ? flags[g] : 0u;
    const bool chain = valid && start[g] == (i ? end[g - 1] : job.first_bit);
    const uint32_t sum = block_sum<kSubBlock / kWave>(valid ? bytes[g] : 0u, red);
    const uint32_t e = block_min<kSubBlock / kWave>((f & kSubEob) ? t : (uint32_t)kSubBlock, red);
    const uint32_t nc = block_min<kSubBlock / kWave>((valid && !chain) ? t : (uint32_t)kSubBlock, red);
    const uint32_t inv = block_min<kSubBlock / kWave>((f & kSubInvalid) ? t : (uint32_t)kSubBlock, red);
    if (t == 0) {
        DecBlockRec r;
        r.sum = sum, r.first_eob = e, r.first_unchained = nc, r.first_invalid = inv;
        recs[blk] = r;
    }
}

// ---- dec_offsets_kernel, one workgroup per file: the stream ends with the FIRST end-of-block symbol of the chain (what lies
//      behind it is padding and the Adler-32, decoded as garbage by their threads); up to there the chain must hold and no
//      subsequence may be invalid; exclusive scan of the blocks' byte counts; the total must be the filtered image ----
__global__ __launch_bounds__(kDecBlock) void dec_offsets_kernel(const DecJob *jobs, const DecBlockRec *recs, const uint32_t *bytes, uint64_t *block_off,
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
    // blocks in front of the last one count whole; of the last one, subsequences 0..last_local
    uint64_t local = 0;
    uint32_t bad = 0;
    for (uint32_t b = i0; b < i1 && b < last_blk; b++) {
        const DecBlockRec r = recs[b0 + b];
        local += r.sum;
        if (r.first_unchained < (uint32_t)kSubBlock) bad |= kDecNotConverged;
        if (r.first_invalid < (uint32_t)kSubBlock) bad |= kDecBadStream;
    }
    uint32_t tail = 0;
    if (last_blk < nb) {
        const DecBlockRec r = recs[b0 + last_blk];
        if (t == 0) {
            if (r.first_unchained <= last_local) bad |= kDecNotConverged;
            if (r.first_invalid <= last_local) bad |= kDecBadStream;
        }
        for (uint32_t k = t; k <= last_local; k += kDecBlock) tail += bytes[job.sub_base + last_blk * kSubBlock + k];
    } else if (t == 0)
        bad |= kDecBadStream;
    const uint32_t tail_sum = block_sum<kDecBlock / kWave>(tail, red);
    const uint32_t any_bad = block_sum<kDecBlock / kWave>(bad & kDecNotConverged, red) ? kDecNotConverged : 0u;
    const uint32_t any_bad2 = block_sum<kDecBlock / kWave>(bad & kDecBadStream, red) ? kDecBadStream : 0u;
    sums[t] = local;
    __syncthreads();
    if (t == 0) {
        uint64_t acc = 0;
        for (int k = 0; k < kDecBlock; k++) {
            const uint64_t v = sums[k];
            sums[k] = acc;
            acc += v;
        }
        uint32_t st = any_bad | any_bad2;
        if (acc + tail_sum != (uint64_t)(job.bpl + 1) * job.h) st |= kDecBadStream; // too few or too many pixels
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

// ---- the real decode ----
__global__ __launch_bounds__(kSubBlock) void dec_emit_kernel(const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t total_subs, const uint64_t *start,
                                                             const uint32_t *bytes, const uint32_t *eob_index, const uint64_t *block_off, uint32_t *status)
{
    __shared__ __attribute__((aligned(16))) uint32_t lut[4096];
    __shared__ uint32_t bits[kSliceSlots];
    __shared__ uint32_t wsum[kSubBlock / kWave];
    const uint32_t blk = first_block + blockIdx.x, g0 = blk * kSubBlock;
    if (g0 >= total_subs) return;
    uint32_t local0;
    const DecJob &job = job_of_sub(jobs, n_jobs, g0, local0);
    const uint32_t job_index = (uint32_t)(&job - jobs);
    if (status[job_index] & ~kDecSawEob) return; // (uniform per block: one file per block)
    const uint32_t last = eob_index[job_index];
    if (local0 > last) return; // behind the end of the stream
    const uint64_t base = stage_block(job, local0, lut, bits);
    const uint32_t g = g0 + threadIdx.x, i = local0 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool active = i < job.n_sub && i <= last;
    // where this thread writes: the block's offset + the byte counts of the block's threads in front of it
    const uint32_t nb = active ? bytes[g] : 0u;
    uint32_t incl = nb;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, kWave);
        if ((int)lane >= o) incl += up;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads(); // (also: the staged tables and bits are there)
    uint32_t before = incl - nb;
    for (uint32_t q = 0; q < wv; q++) before += wsum[q];
    if (!active) return;
    const uint32_t boundary = (uint32_t)(job.first_bit + (uint64_t)(i + 1) * kSubBits - base);
    const uint64_t total = (uint64_t)(job.bpl + 1) * job.h;
    const uint32_t stride = job.bpl + 1, c = job.src_c, wpr = (job.w + 31) >> 5;
    LdsBits in;
    in.l = bits;
    const uint64_t lim = job.end_limit_bit - base;
    in.limit = lim > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)lim;
    in.pos = (uint32_t)(start[g] - base);
    const uint64_t o0 = block_off[blk] + before; // stream byte this thread starts at
    uint32_t row = (uint32_t)(o0 / stride), col = (uint32_t)(o0 - (uint64_t)row * stride);
    // Everything inside the loop is 32-bit and relative to the thread's start: its tokens cover at most kSubBits / 2 matches of
    // 258 bytes.  `left` = stream bytes the image still takes; a = byte offset from Fal (the thread's first buffer byte rounded
    // down to a dword: buffer byte of stream position (row, col) = row * fstride + 3 + col).
    if (o0 > total) { // (cannot happen behind dec_offsets_kernel's total check; kept as a guard)
        atomicOr(&status[job_index], kDecBadStream);
        return;
    }
    const uint64_t left64 = total - o0;
    const uint32_t left = left64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)left64;
    uint32_t o = 0; // stream bytes this thread has produced
    const size_t a0 = (size_t)row * job.fstride + 3u + col;
    FPNG_DEC_GLOBAL uint8_t *Fal = (FPNG_DEC_GLOBAL uint8_t *)(uintptr_t)(job.filt + (a0 & ~(size_t)3));
    FPNG_DEC_GLOBAL uint32_t *mask_row = (FPNG_DEC_GLOBAL uint32_t *)(uintptr_t)(job.runmask + (size_t)row * wpr);
    uint32_t a = (uint32_t)(a0 & 3u);
    const uint32_t row_gap = job.fstride - stride;
    // literals are collected into the dword they fall in and stored with one instruction when all four of its bytes are
    // literals of THIS thread; bytes of run pixels are dec_fill_kernel's, a dword shared with the neighbouring thread or broken
    // by a run is stored byte by byte
    uint32_t cur = ~0u; // dword (a / 4) being collected
    uint32_t acc = 0, have = 0;
    auto flush = [&]() {
        if (have == 0xFu)
            *(FPNG_DEC_GLOBAL uint32_t *)(Fal + cur * 4u) = acc;
        else
            for (uint32_t k = 0; k < 4; k++)
                if (have & (1u << k)) Fal[cur * 4u + k] = (uint8_t)(acc >> (8 * k));
        have = 0, acc = 0;
    };
    // pixels per byte count: c is 3 or 4 (x / 3 by multiplication: x < 2^31)
    auto div_c = [&](uint32_t x) { return c == 4 ? x >> 2 : (uint32_t)(((uint64_t)x * 0xAAAAAAABull) >> 33); };
    uint32_t err = 0;
    while (in.pos < boundary) {
        if (in.pos >= in.limit) {
            err = kDecBadStream;
            break;
        }
        uint32_t run = 0;
        const int t = next_token(in, lut, run);
        if (t < 0) {
            err = kDecBadStream;
            break;
        }
        if (t == 256) { // end of block: every pixel must be there, and the stream must end 4 bytes (the Adler-32) before the IDAT does
            if (o != left || left64 > 0x7FFFFFFFull || ((base + in.pos + 7) >> 3) + 4 != job.z_bytes) err = kDecBadStream;
            atomicOr(&status[job_index], kDecSawEob);
            break;
        }
        if (t < 256) {
            if (o >= left || (col == 0 && (uint32_t)t != (row ? 2u : 0u))) { // the row's filter literal: 0, then 2 (Up)
                err = kDecBadStream;
                break;
            }
            if (col) { // (the filter literal itself is not kept)
                if ((a >> 2) != cur) {
                    if (have) flush();
                    cur = a >> 2;
                }
                acc |= (uint32_t)t << (8 * (a & 3)), have |= 1u << (a & 3);
            }
            o++, a++;
            if (++col == stride) col = 0, row++, a += row_gap, mask_row += wpr;
        } else {
            // a match repeats the previous pixel: whole pixels, inside the row (reference fpng.cpp:2301-2330)
            const uint32_t x = div_c(col - 1), npix = div_c(run);
            if (col == 0 || x * c != col - 1 || npix * c != run || !npix || x + npix > job.w || run > left - o) {
                err = kDecBadStream;
                break;
            }
            for (uint32_t p = x; p < x + npix;) { // set bits [x, x + npix)
                const uint32_t wd = p >> 5, b0 = p & 31, cnt = min(32u - b0, x + npix - p);
                atomicOr((uint32_t *)(uintptr_t)&mask_row[wd], (cnt == 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u)) << b0);
                p += cnt;
            }
            o += run, a += run;
            col += run;
            if (col == stride) col = 0, row++, a += row_gap, mask_row += wpr;
        }
    }
    if (have) flush();
    if (err) atomicOr(&status[job_index], err);
}

// ---- runs: every marked pixel takes the filtered value of the nearest unmarked pixel to its left (zero if there is none) ----
__global__ __launch_bounds__(kDecBlock) void dec_fill_kernel(const DecJob *jobs, const uint32_t *status)
{
    const DecJob &job = jobs[blockIdx.y];
    if (job.mode != 0 || (status[blockIdx.y] & ~kDecSawEob)) return;
    const uint32_t lane = threadIdx.x & 63, row = blockIdx.x * (kDecBlock / kWave) + dec_uniform(threadIdx.x >> 6);
    if (row >= job.h) return;
    const uint32_t c = job.src_c, wpr = (job.w + 31) >> 5;
    uint8_t *F = job.filt + (size_t)row * job.fstride + 4; // (4-byte aligned: 4-channel pixels are dwords)
    const uint32_t *m = job.runmask + (size_t)row * wpr;
    auto load_px = [&](uint32_t x) -> uint32_t {
        if (c == 4) return *(const uint32_t *)(F + (size_t)x * 4);
        const uint8_t *p = F + (size_t)x * 3;
        return p[0] | (p[1] << 8) | (p[2] << 16);
    };
    uint32_t carry = 0; // value of the last pixel of the previous window (filtered bytes, packed)
    for (uint32_t x0 = 0; x0 < job.w; x0 += 64) {
        const uint32_t x = x0 + lane;
        const bool valid = x < job.w;
        const bool is_run = valid && ((m[x >> 5] >> (x & 31)) & 1);
        const uint64_t runs = __ballot(is_run);
        const uint32_t last = min(63u, job.w - 1 - x0); // lane of the window's last pixel
        if (!runs) { // nothing to fill in this window: only its last pixel matters (to the next one)
            const uint32_t v = (lane == last) ? load_px(x) : 0u;
            carry = (uint32_t)__shfl((int)v, (int)last, kWave);
            continue;
        }
        const uint32_t v = (valid && !is_run) ? load_px(x) : 0u;
        const uint64_t lit = __ballot(valid && !is_run);
        const uint64_t le = (2ull << lane) - 1ull;
        const uint64_t below = lit & le;
        const int src = below ? 63 - __builtin_clzll(below) : -1; // nearest literal pixel at or below this lane
        const uint32_t got = (uint32_t)__shfl((int)v, src < 0 ? 0 : src, kWave);
        const uint32_t val = src < 0 ? carry : got;
        if (is_run) {
            if (c == 4)
                *(uint32_t *)(F + (size_t)x * 4) = val;
            else {
                uint8_t *p = F + (size_t)x * 3;
                p[0] = (uint8_t)val, p[1] = (uint8_t)(val >> 8), p[2] = (uint8_t)(val >> 16);
            }
        }
        carry = (uint32_t)__shfl((int)val, (int)last, kWave);
    }
}

// ---- Up filter undone: out[y] = out[y-1] + filtered[y] (bytes, mod 256); one thread per DWORD column (four byte columns: packed
//      byte adds) and SEGMENT of kUnfRows rows -- a column alone is a chain of h dependent steps and an 8K frame has only 7680 of
//      them.  dec_unfilter_sums_kernel adds up every segment, dec_unfilter_kernel starts from the sum of the segments above its
//      own (<= h / kUnfRows loads) and writes the pixels, 3 <-> 4 channels on the way out ----
constexpr uint32_t kUnfRows = kDecUnfRows;
__device__ __forceinline__ uint32_t add_bytes(uint32_t a, uint32_t b)
{
    return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u);
}
__global__ __launch_bounds__(kDecBlock) void dec_unfilter_sums_kernel(const DecJob *jobs, const uint32_t *status)
{
    const DecJob &job = jobs[blockIdx.z];
    if (job.mode != 0 || (status[blockIdx.z] & ~kDecSawEob)) return;
    const uint32_t j4 = blockIdx.x * kDecBlock + threadIdx.x, sg = blockIdx.y, ncol = job.fstride / 4 - 1;
    if (j4 >= ncol || sg + 1 >= job.nseg) return; // (nobody reads the last segment's sum)
    const uint32_t *F = (const uint32_t *)(job.filt + 4) + j4;
    const size_t fs4 = job.fstride / 4;
    const uint32_t y0 = sg * kUnfRows;
    uint32_t acc = 0;
#pragma unroll 8
    for (uint32_t y = y0; y < y0 + kUnfRows; y++) acc = add_bytes(acc, F[(size_t)y * fs4]);
    job.segsum[(size_t)sg * ncol + j4] = acc;
}
__global__ __launch_bounds__(kDecBlock) void dec_unfilter_kernel(const DecJob *jobs, const uint32_t *status)
{
    const DecJob &job = jobs[blockIdx.z];
    if (job.mode != 0 || (status[blockIdx.z] & ~kDecSawEob)) return;
    const uint32_t j4 = blockIdx.x * kDecBlock + threadIdx.x, sg = blockIdx.y; // dword column of the file's rows, segment of rows
    if (j4 * 4 >= job.bpl || sg >= job.nseg) return;
    const uint32_t sc = job.src_c, dc = job.dst_c, nb = min(4u, job.bpl - j4 * 4), ncol = job.fstride / 4 - 1;
    const uint32_t *F = (const uint32_t *)(job.filt + 4) + j4;
    const size_t fs4 = job.fstride / 4, os = (size_t)job.w * dc;
    const bool whole = sc == dc && nb == 4 && (os & 3) == 0 && (((uintptr_t)job.out) & 3) == 0; // aligned dword stores
    uint32_t acc = 0;
    for (uint32_t q = 0; q < sg; q++) acc = add_bytes(acc, job.segsum[(size_t)q * ncol + j4]);
    const uint32_t y0 = sg * kUnfRows, y1 = min(job.h, y0 + kUnfRows);
    for (uint32_t y = y0; y < y1; y++) {
        acc = add_bytes(acc, F[(size_t)y * fs4]);
        uint8_t *orow = job.out + (size_t)y * os;
        if (whole)
            *(uint32_t *)(orow + (size_t)j4 * 4) = acc;
        else
            for (uint32_t k = 0; k < nb; k++) {
                const uint32_t j = j4 * 4 + k, px = j / sc, ch = j - px * sc;
                if (ch >= dc) continue; // alpha dropped
                orow[(size_t)px * dc + ch] = (uint8_t)(acc >> (8 * k));
                if (dc == 4 && sc == 3 && ch == 2) orow[(size_t)px * dc + 3] = 0xFF;
            }
    }
}

// ---- stored files: the filter-0 stream sits in stored blocks of 65535 bytes (the host checked their headers and the filter bytes) ----
__global__ __launch_bounds__(kDecBlock) void dec_stored_kernel(const DecJob *jobs)
{
    const DecJob &job = jobs[blockIdx.y];
    if (job.mode != 1) return;
    const uint64_t n = (uint64_t)job.w * job.h * job.dst_c;
    for (uint64_t k = (uint64_t)blockIdx.x * kDecBlock + threadIdx.x; k < n; k += (uint64_t)gridDim.x * kDecBlock) {
        const uint64_t pixel = k / job.dst_c;
        const uint32_t ch = (uint32_t)(k - pixel * job.dst_c);
        uint8_t v = 0xFF;
        if (ch < job.src_c) {
            const uint64_t y = pixel / job.w, x = pixel - y * job.w;
            const uint64_t s = y * (job.bpl + 1) + 1 + x * job.src_c + ch; // stream byte
            v = job.z[2 + 5 * (s / 65535 + 1) + s];
        }
        job.out[k] = v;
    }
}

} // namespace

// (the decoding kernels work on the workgroups [first_block, first_block + n_blocks) of the batch's subsequences: one group of files)
void launch_dec_sync(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, uint32_t round, uint64_t *start,
                     uint64_t *end, uint32_t *bytes, uint32_t *flags, uint32_t *changed)
{
    hipLaunchKernelGGL(dec_sync_kernel, dim3(n_blocks), dim3(kSubBlock), 0, s, jobs, n_jobs, first_block, total_subs, round, start, end, bytes, flags, changed);
}
// group_jobs / status / eob_index: of the group's first file
void launch_dec_offsets(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, const DecJob *group_jobs,
                        uint32_t n_group_jobs, const uint64_t *start, const uint64_t *end, const uint32_t *bytes, const uint32_t *flags, DecBlockRec *recs,
                        uint64_t *block_off, uint32_t *status, uint32_t *eob_index)
{
    hipLaunchKernelGGL(dec_blocksum_kernel, dim3(n_blocks), dim3(kSubBlock), 0, s, jobs, n_jobs, first_block, total_subs, start, end, bytes, flags, recs);
    hipLaunchKernelGGL(dec_offsets_kernel, dim3(n_group_jobs), dim3(kDecBlock), 0, s, group_jobs, recs, bytes, block_off, status, eob_index);
}
// status / eob_index: of the batch's first file
void launch_dec_emit(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, const uint64_t *start,
                     const uint32_t *bytes, const uint32_t *eob_index, const uint64_t *block_off, uint32_t *status)
{
    hipLaunchKernelGGL(dec_emit_kernel, dim3(n_blocks), dim3(kSubBlock), 0, s, jobs, n_jobs, first_block, total_subs, start, bytes, eob_index, block_off, status);
}
void launch_dec_finish(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t max_rows, uint32_t max_bpl, const uint32_t *status)
{
    const uint32_t rows_per_block = kDecBlock / kWave;
    hipLaunchKernelGGL(dec_fill_kernel, dim3((max_rows + rows_per_block - 1) / rows_per_block, n_jobs), dim3(kDecBlock), 0, s, jobs, status);
    const dim3 ugrid(((max_bpl + 3) / 4 + kDecBlock - 1) / kDecBlock, (max_rows + kUnfRows - 1) / kUnfRows, n_jobs);
    if (ugrid.y > 1) hipLaunchKernelGGL(dec_unfilter_sums_kernel, ugrid, dim3(kDecBlock), 0, s, jobs, status);
    hipLaunchKernelGGL(dec_unfilter_kernel, ugrid, dim3(kDecBlock), 0, s, jobs, status);
    hipLaunchKernelGGL(dec_stored_kernel, dim3(1024, n_jobs), dim3(kDecBlock), 0, s, jobs);
}

} // namespace fpng_amd
