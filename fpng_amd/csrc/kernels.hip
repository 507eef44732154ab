// kernels.hip -- CDNA4 (gfx950, wave64) kernels of the fpng encode hot path.
//
// Whole images, per submission (one internal stream, no host round trip):
//
//   [hist_kernel, build_dynamic_kernel]   2-pass only: symbol histogram, per-image Huffman table + block header
//   encode_rows_kernel<C>  one wavefront per scanline, ONE walk over the pixels: Up/None filter on the fly, RLE
//                    chunking from wave ballots, tokens from LDS tables, DPP prefix sum of token lengths -> the
//                    row's private, dword-aligned LOCAL STREAM (LDS staging window, coalesced stores); bits per
//                    row and Adler-32 partial sums on the side.  (reference fpng.cpp:1592-1660 filter,
//                    :1468-1558 / :1182-1241 token grammar, :407-487 Adler)
//   scan_kernel      per image: exclusive scan of row bits -> absolute bit offset of every row, Adler combine,
//                    closed-form "did the coder run out of buffer" decision (reference fpng.cpp:567-588),
//                    PNG header + Deflate prefix.
//   assemble_kernel  shifts the local streams to their bit positions (rows meet inside a dword), stores the
//                    file 16 bytes per lane and takes the CRC-32 of the same bytes: slice-by-16 from LDS, lane
//                    stripes folded with GF(2) constants (reference fpng.cpp:234-292).  Images that fell back get
//                    their stored blocks here instead (reference fpng.cpp:818-866), straight from the pixels.
//   finalize_kernel  folds the CRC partials, writes Adler / IDAT CRC / IEND and the result record
//                    (reference fpng.cpp:1764-1800).
//
// Row bands (multi-GPU single image) use the same kernels: encode_rows_kernel + scan_kernel in counting mode, then
// scan_kernel + assemble_kernel into a window of the file (bits of other bands stay zero); crc_kernel is the CRC pass of
// fpng_amd_wrap_png() for callers that do not hand over the bands' CRC partials.
//
// Integer / byte work throughout, bounded by VALU issue and HBM traffic: no MFMA.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

namespace fpng_amd {

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;
// Rows (waves) per block of encode_rows_kernel.  Every block re-reads the row above its first row (another block's row,
// rarely still in L2), so bigger blocks mean less traffic -- but coarser scheduling costs more: measured on one box,
// 8 x 8K RGBA / 256 x 1080p RGB / 1024 x 512^2: 16 rows 480 / 465 / - GP/s, 8 rows 517 / 482 / 314, 4 rows 537 / 522 / 323,
// 3 rows 524 / 503 / 301, 2 rows 513 / 501 / 304.
#ifndef FPNG_ROW_WAVES
#define FPNG_ROW_WAVES 4
#endif
constexpr int kRowWaves = FPNG_ROW_WAVES;
constexpr int kScanWaves = 4;              // scan_kernel: one block per job, its threads split the rows
constexpr int kScanBlock = kWave * kScanWaves;
constexpr int kHistWaves = 8;              // hist_kernel: one LDS histogram (36 KiB) and one round of global atomics per block
constexpr int kHistBlock = kWave * kHistWaves;
constexpr int kRowBlock = kWave * kRowWaves;
#ifndef FPNG_STAGE_DWORDS
#define FPNG_STAGE_DWORDS 1024
#endif
#ifndef FPNG_ROWS_WPE // waves per SIMD the row kernels are compiled for (build variants: occupancy against window size)
#define FPNG_ROWS_WPE 8
#endif
// ... and the 4-channel one on WIDE rows (launch_encode_rows' `wide4`: most of the batch's pixels lie in rows of kWideRowPixels and more):
// SEVEN.  Same-box A/Bs on two boxes (profiles/r05_rows_w6.txt): on long rows the 4-channel walk is as fast with six or seven waves per SIMD
// as with eight (no register spills), and the registers and LDS it leaves free let the other lane's kernels -- the next submission's
// histogram pass, the previous one's assemble -- run NEXT to it instead of behind it: 8 x 8K 2-pass + 2.5 ... 4.4 % with six or seven,
// 1-pass + 1.7 % / - 0.5 % with six (by box), + 0.8 % with seven; five waves lose 2 %.  Short rows need their eight waves (256 x 1080p
// RGBA - 5.5 % with six), and so does the 3-channel walk (- 10 %).
#ifndef FPNG_ROWS_WPE4
#define FPNG_ROWS_WPE4 7
#endif
constexpr int kStageDwords = FPNG_STAGE_DWORDS; // per-wave LDS staging window of the output bit stream
// Local-stream stores carry the non-temporal hint (build with -DFPNG_LOCAL_NT=0 to A/B it: the hint decides whether the
// streams are kept in L2 / Infinity Cache for assemble_kernel)
#ifndef FPNG_LOCAL_NT
#define FPNG_LOCAL_NT 1
#endif
template <typename T, typename P> __device__ __forceinline__ void local_store(const T &v, P p)
{
#if FPNG_LOCAL_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
constexpr int kStageFlushAt = kStageDwords - 136; // a 64-pixel window adds at most 64*60 bits = 120 dwords

// ---------------------------------------------------------------------------------------------
// wave-level primitives (wave64, gfx9 DPP)
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND_CTRL = true>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL);
}

// value of lane-1; lane 0 receives `lane0_value`
__device__ __forceinline__ uint32_t lane_prev(uint32_t v, uint32_t lane0_value)
{
    return dpp_mov<0x138, 0xF, 0xF, false>(lane0_value, v); // wave_shr:1
}

// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    v += dpp_mov<0x111>(0, v);             // row_shr:1
    v += dpp_mov<0x112>(0, v);             // row_shr:2
    v += dpp_mov<0x114>(0, v);             // row_shr:4
    v += dpp_mov<0x118>(0, v);             // row_shr:8   -> scan inside each row of 16
    v += dpp_mov<0x142, 0xA>(0, v);        // row_bcast:15 into rows 1,3
    v += dpp_mov<0x143, 0xC>(0, v);        // row_bcast:31 into rows 2,3
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_sum(v), 63);
}

__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)uniform((uint32_t)(v >> 32)) << 32) | uniform((uint32_t)v);
}

// LDS traffic between lanes of ONE wave: the hardware keeps a wave's DS operations in order; this
// only stops the compiler from moving them across.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Pointers that come out of a Job record are generic ("flat") to the compiler.  The hot loads and
// stores go through explicitly global-address-space pointer types so they become global_*
// instructions (flat_* ones also occupy the LDS counter and issue slower).
#define FPNG_GLOBAL __attribute__((address_space(1)))
typedef const FPNG_GLOBAL uint8_t *gptr_cu8;
typedef const FPNG_GLOBAL uint32_t *gptr_cu32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const FPNG_GLOBAL u32x4 *gptr_cu128;
typedef FPNG_GLOBAL u32x4 *gptr_u128;
typedef FPNG_GLOBAL uint8_t *gptr_u8;
typedef FPNG_GLOBAL uint32_t *gptr_u32;
template <typename G, typename T> __device__ __forceinline__ G to_global(T *p) { return (G)(uintptr_t)p; }

// per-byte a-b (mod 256) on four packed bytes
__device__ __forceinline__ uint32_t sub_bytes(uint32_t a, uint32_t b)
{
    return ((a | 0x80808080u) - (b & 0x7F7F7F7Fu)) ^ ((a ^ ~b) & 0x80808080u);
}

// the same for four dwords with SDWA byte operands: 4 instructions per dword instead of 6; the four dwords are
// interleaved so that the read-modify-write of a destination never follows its predecessor directly
__device__ __forceinline__ void sub_bytes_x4(const u32x4 &a, const u32x4 &b, uint32_t (&r)[4])
{
    uint32_t r0, r1, r2, r3;
    const uint32_t a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
#define FPNG_SUB_BYTE0(R, A, B) \
    asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(R) : "v"(A), "v"(B))
#define FPNG_SUB_BYTE(K, R, A, B)                                                                                        \
    asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:BYTE_" #K " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #K " src1_sel:BYTE_" #K \
        : "+v"(R)                                                                                                        \
        : "v"(A), "v"(B))
    FPNG_SUB_BYTE0(r0, a0, b0); FPNG_SUB_BYTE0(r1, a1, b1); FPNG_SUB_BYTE0(r2, a2, b2); FPNG_SUB_BYTE0(r3, a3, b3);
    FPNG_SUB_BYTE(1, r0, a0, b0); FPNG_SUB_BYTE(1, r1, a1, b1); FPNG_SUB_BYTE(1, r2, a2, b2); FPNG_SUB_BYTE(1, r3, a3, b3);
    FPNG_SUB_BYTE(2, r0, a0, b0); FPNG_SUB_BYTE(2, r1, a1, b1); FPNG_SUB_BYTE(2, r2, a2, b2); FPNG_SUB_BYTE(2, r3, a3, b3);
    FPNG_SUB_BYTE(3, r0, a0, b0); FPNG_SUB_BYTE(3, r1, a1, b1); FPNG_SUB_BYTE(3, r2, a2, b2); FPNG_SUB_BYTE(3, r3, a3, b3);
#undef FPNG_SUB_BYTE0
#undef FPNG_SUB_BYTE
    r[0] = r0, r[1] = r1, r[2] = r2, r[3] = r3;
}

// ---------------------------------------------------------------------------------------------
// RLE structure of one 64-pixel window from the wave ballot of "equals previous pixel".
//
// A run of equal pixels is cut greedily from its first pixel into chunks of at most CAP pixels
// (reference fpng.cpp:1503-1514 / :1209-1219).  We attribute every chunk to its LAST pixel:
//   t(x)  = number of consecutive "same" pixels ending at x
//   q(x)  = ((t-1) mod CAP) + 1        position inside the current chunk
//   x ends a chunk iff same(x) and (q == CAP or !same(x+1))
// which needs only look-back (a scalar carry across windows) and one pixel of look-ahead.
// ---------------------------------------------------------------------------------------------
template <int C> struct Rle {
    static constexpr uint32_t CAP = (C == 4) ? kMaxChunkPixels4 : kMaxChunkPixels3;
    uint32_t carry = 0; // t of the pixel just before the window, reduced mod CAP (wave-uniform)

    // returns q (1..CAP) for lanes with same==1 (undefined otherwise); ends = chunk-end flag
    __device__ __forceinline__ uint32_t classify(uint64_t same_mask, uint32_t next_same_bit0, uint32_t lane,
                                                 uint64_t lane_le_mask, bool &ends) const
    {
        const uint64_t zeros_le = ~same_mask & lane_le_mask; // not-same pixels at or below this lane
        uint32_t t;
        if (zeros_le == 0)
            t = carry + lane + 1;
        else
            t = lane - (63u - (uint32_t)__builtin_clzll(zeros_le));
        uint32_t u = t - 1;
        if (u >= CAP) u -= CAP;
        if (C == 4 && u >= CAP) u -= CAP;
        const uint32_t q = u + 1;
        const uint64_t after = (same_mask >> 1) | ((uint64_t)next_same_bit0 << 63);
        const bool same = (same_mask >> lane) & 1;
        ends = same && (q == CAP || !((after >> lane) & 1));
        return q;
    }

    // advance the carry past a full window
    __device__ __forceinline__ void advance(uint64_t same_mask)
    {
        uint32_t t;
        if (same_mask == ~0ull)
            t = carry + 64;
        else
            t = (uint32_t)__builtin_clzll(~same_mask); // leading ones
        while (t >= CAP) t -= CAP;
        carry = t;
    }
};

// LDS table layout of the row walkers.  lit[s]: code in bits 0..15, code length in bits 24..31 (field
// extraction folds into SDWA operand selects; four entries can be added and the top byte is the sum of
// the lengths).  chunk[q]: total bit count in bits 0..7, token bits from bit 8.
struct PackedTables {
    uint32_t lit[288];
    uint32_t chunk[96];
};

template <int BLOCK> __device__ __forceinline__ void stage_packed_tables(PackedTables &dst, const TokenTable *src)
{
    // trip counts known at compile time: all of a thread's loads are issued before the first one is needed
    uint32_t e[(288 + BLOCK - 1) / BLOCK], ch = 0;
#pragma unroll
    for (int k = 0; k < (288 + BLOCK - 1) / BLOCK; k++) {
        const int i = (int)threadIdx.x + k * BLOCK;
        e[k] = i < 288 ? src->lit[i] : 0u;
    }
    static_assert(BLOCK >= 96, "one round for the chunk table");
    if (threadIdx.x < 96) ch = src->chunk[threadIdx.x];
#pragma unroll
    for (int k = 0; k < (288 + BLOCK - 1) / BLOCK; k++) {
        const int i = (int)threadIdx.x + k * BLOCK;
        if (i < 288) dst.lit[i] = (e[k] & 0xFFFFu) | ((e[k] >> 16) << 24);
    }
    if (threadIdx.x < 96) dst.chunk[threadIdx.x] = (ch >> 24) | ((ch & 0xFFFFFFu) << 8);
}

// packed literal entry: code in the low word, length in the top byte (both are SDWA operand selects)
__device__ __forceinline__ uint32_t plit_code(uint32_t e) { return e & 0xFFFFu; }
__device__ __forceinline__ uint32_t plit_len(uint32_t e) { return e >> 24; }

// Two packed entries a (first in the stream), b: (code(b) << len(a)) | code(a), and len(a) + len(b).  Written with
// SDWA operand selects by hand: the compiler picks them for the shift and the length sum, but masks the codes with
// separate v_and instructions and builds the upper half with VOP3 forms that cannot select sub-dwords (11 instructions
// per RGBA pixel instead of 9; this is the innermost arithmetic of the walk).
__device__ __forceinline__ uint32_t join_codes(uint32_t a, uint32_t b)
{
    uint32_t t, r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:WORD_0" : "=v"(t) : "v"(a), "v"(b));
    asm("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(t), "v"(a));
    return r;
}
__device__ __forceinline__ uint32_t add_lens(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// all literals of one pixel as one token; entries are packed (see PackedTables)
template <int C> __device__ __forceinline__ uint64_t packed_literal_token(const PackedTables &T, uint32_t f, uint32_t &nbits)
{
    const uint32_t e0 = T.lit[f & 0xFF], e1 = T.lit[(f >> 8) & 0xFF], e2 = T.lit[(f >> 16) & 0xFF];
    const uint32_t lo = join_codes(e0, e1);
    const uint32_t s01 = add_lens(e0, e1);
    uint32_t hi;
    if (C == 4) {
        const uint32_t e3 = T.lit[f >> 24];
        hi = join_codes(e2, e3);
        nbits = s01 + add_lens(e2, e3);
    } else {
        hi = plit_code(e2);
        nbits = s01 + plit_len(e2);
    }
    uint64_t tok = (uint64_t)hi << s01; // (its low s01 <= 24 bits are zero, lo fits below them)
    return tok | lo;
}

// ---- pixel windows through buffer resources: out-of-range lanes read 0, no exec masking ----
// The descriptor must live in SGPRs; whenever the compiler cannot prove base/size wave-uniform it wraps
// EVERY load through it in a waterfall loop (cdna_hip_programming.md T20), so force them uniform here.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    const uint64_t p = uniform64((uint64_t)(uintptr_t)base);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)p, 0, (int)uniform(bytes), 0x00020000);
}

template <int C> struct RowWindows {
    __amdgpu_buffer_rsrc_t cur, up;
    uint32_t voff; // per-lane byte offset of its pixel inside window 0 of the row (RGB: aligned down)
    uint32_t sh;   // RGB: byte phase of the pixel inside its aligned dword

    __device__ __forceinline__ void init(const uint8_t *row, const uint8_t *up_row, uint32_t bpl, uint32_t lane)
    {
        if (C == 4) {
            cur = make_rsrc(row, bpl);
            up = make_rsrc(up_row ? up_row : row, up_row ? bpl : 0);
            voff = lane * 4;
            sh = 0;
        } else {
            // base pointers aligned down to 4 so that every load is an aligned dword; the row then
            // starts `a` bytes into the resource.  Up row has its own phase.
            const uint32_t a = (uint32_t)((uintptr_t)row & 3);
            phase = uniform(a);
            cur = make_rsrc(row - a, (a + bpl + 3) & ~3u);
            voff = (3 * lane + a) & ~3u;
            sh = (3 * lane + a) & 3u;
            // branch-free: a descriptor chosen by a branch ends up in VGPRs and every load through it
            // in a waterfall loop.  No Up row -> zero-sized resource (reads return 0).
            const uint8_t *ub = up_row ? up_row : row;
            const uint32_t b = (uint32_t)((uintptr_t)ub & 3);
            up_phase = uniform(b);
            up = make_rsrc(ub - b, up_row ? ((b + bpl + 3) & ~3u) : 0u);
            up_voff = (3 * lane + b) & ~3u;
            up_sh = (3 * lane + b) & 3u;
        }
    }
    uint32_t up_voff = 0, up_sh = 0;
    uint32_t phase = 0, up_phase = 0; // RGB: byte offset of the row / Up row inside its (dword-aligned) resource

    // Raw dwords of this lane's pixel in the 64-pixel window starting at pixel x0 (multiple of 64).
    // Loading and filtering are split so that several windows can be in flight (the walk keeps a
    // 4-deep ring of Raw values: HBM latency is covered by the ring, not only by other waves).
    struct Raw {
        uint32_t c_lo, c_hi, u_lo, u_hi;
    };
    __device__ __forceinline__ Raw load_raw(uint32_t x0) const
    {
        Raw q;
        if (C == 4) {
            q.c_lo = __builtin_amdgcn_raw_buffer_load_b32(cur, voff, x0 * 4, 0);
            q.u_lo = __builtin_amdgcn_raw_buffer_load_b32(up, voff, x0 * 4, 0);
            q.c_hi = q.u_hi = 0;
        } else {
            const uint32_t so = x0 * 3; // 192-byte steps keep the dword phase of every lane
            q.c_lo = __builtin_amdgcn_raw_buffer_load_b32(cur, voff, so, 0);
            q.c_hi = __builtin_amdgcn_raw_buffer_load_b32(cur, voff + 4, so, 0);
            q.u_lo = __builtin_amdgcn_raw_buffer_load_b32(up, up_voff, so, 0);
            q.u_hi = __builtin_amdgcn_raw_buffer_load_b32(up, up_voff + 4, so, 0);
        }
        return q;
    }
    // filtered pixel: bytes of (cur - up) mod 256 (reference fpng.cpp:1605-1655)
    __device__ __forceinline__ uint32_t filter(const Raw &q) const
    {
        if (C == 4) return sub_bytes(q.c_lo, q.u_lo);
        const uint32_t c = __builtin_amdgcn_alignbyte(q.c_hi, q.c_lo, sh);
        const uint32_t u = __builtin_amdgcn_alignbyte(q.u_hi, q.u_lo, up_sh);
        return sub_bytes(c, u) & 0xFFFFFFu;
    }
    __device__ __forceinline__ uint32_t filtered_at(uint32_t x0) const { return filter(load_raw(x0)); }
};

// 64-bit mask of lanes whose pixel index x0+lane is below `limit`
__device__ __forceinline__ uint64_t valid_mask(uint32_t x0, uint32_t limit)
{
    if (limit >= x0 + 64) return ~0ull;
    if (limit <= x0) return 0ull;
    return (1ull << (limit - x0)) - 1ull;
}


// ---------------------------------------------------------------------------------------------
// Row walker shared by the count / histogram / emit kernels: one wavefront walks one scanline in
// 64-pixel windows.
//
// Instruction count is what bounds these kernels (VALU issue, not HBM), so the walk is organised
// around the common cases:
//   * raw pixel loads go through buffer resources (hardware bounds check, SGPR base + lane offset)
//     and a 4-deep register ring, so nothing in the loop waits for memory it just asked for;
//   * interior windows (this one and the next are completely inside the row) carry no validity
//     masks at all; only the last two windows of a row run the masked "tail" body;
//   * a window whose `same` mask is zero (no pixel repeats its left neighbour) skips the whole RLE
//     classification: every lane is a literal pixel;
//   * table entries are packed so that code lengths add up in the low byte and an entry can be used
//     directly as a shift amount.
// ---------------------------------------------------------------------------------------------
// Hist: symbol histogram (2-pass, pass 1).  Encode: tokens into the wave's private stream (a row's scratch stream or a
// segment's LDS window) together with the row's bit count and Adler sums, so that an image needs one walk over its pixels.
enum class Pass { Hist, Encode };

struct EmitSink {
    uint32_t *stage;     // this wave's LDS window
    gptr_u32 out32;      // dword view of the destination (bit 0 of dword 0 = destination bit 0)
    uint64_t base_dw;    // destination dword index of stage[0]
    uint32_t fill;       // bits used in the window
    bool wide;           // flushed 16 bytes per lane (sink_flush_exclusive)
};

__device__ __forceinline__ void sink_zero(EmitSink &s, uint32_t lane, uint32_t ndw)
{
    for (uint32_t j = lane; j < ndw; j += kWave) s.stage[j] = 0;
}
// the whole window, once per row: four ds_write_b128 per lane instead of sixteen ds_write_b32 (the dump slots behind the window only
// ever receive zeros)
__device__ __forceinline__ void sink_zero_window(EmitSink &s, uint32_t lane)
{
    static_assert(kStageDwords % (4 * kWave) == 0, "whole 16-byte groups per lane");
    u32x4 *st4 = (u32x4 *)s.stage;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < kStageDwords / (4 * kWave); k++) st4[lane + k * kWave] = zero4;
}

// OR a token of nbits (<= 60) at window bit position pos.  Lanes without a token pass code == 0:
// two unconditional ds_or are cheaper than the exec juggling of conditional ones.
// ALL_TOKENS: every lane is known to carry a token (all-literal window), no dump slots needed.
template <bool ALL_TOKENS>
__device__ __forceinline__ void sink_put(EmitSink &s, uint64_t code, uint32_t nbits, uint32_t pos)
{
    // Token-less lanes all share one bit position; atomics from many lanes to ONE LDS address
    // serialise (64-way in run-length-heavy windows), so each of them ORs its zero into a private
    // dump slot behind the window instead.
    uint32_t d = pos >> 5;
    const uint32_t sh = pos & 31;
    if (!ALL_TOKENS) d = nbits ? d : ((uint32_t)kStageDwords + 2u * (threadIdx.x & 63));
    const uint64_t v = code << sh;
    atomicOr(&s.stage[d], (uint32_t)v);
    atomicOr(&s.stage[d + 1], (uint32_t)(v >> 32));
    if (__ballot(sh + nbits > 64)) atomicOr(&s.stage[d + 2], (uint32_t)((code >> 1) >> (63 - sh)));
}

// a token of up to 64 bits (two pixels' literals merged in the lane); the third dword is common here
__device__ __forceinline__ void sink_put_wide(EmitSink &s, uint64_t code, uint32_t pos)
{
    const uint32_t d = pos >> 5, sh = pos & 31;
    const uint64_t v = code << sh;
    atomicOr(&s.stage[d], (uint32_t)v);
    atomicOr(&s.stage[d + 1], (uint32_t)(v >> 32));
    atomicOr(&s.stage[d + 2], (uint32_t)((code >> 1) >> (63 - sh)));
}

// Write out the complete dwords of the window (all of them when `final`), keep the partial one.  The destination is
// the wave's own stream (a row's local stream, a segment's spill area): plain coalesced stores.
// 4-channel local streams: 16 bytes per lane per step -- ds_read_b128, global_store_dwordx4 and the
// re-zeroing ds_write_b128 move four dwords where the generic path below moves one.  Only multiples of four dwords
// leave the window, so the destination stays 16-byte aligned; the final flush rounds up into the row's slack.
__device__ __forceinline__ void sink_flush_exclusive(EmitSink &s, uint32_t lane, bool final)
{
    wave_lds_fence();
    // groups of four dwords to write out: everything at the end of the row, whole 128-byte lines in between (the rows'
    // streams start on line boundaries, so no line of a stream is written in two pieces)
    const uint32_t n4 = final ? ((((s.fill + 31) >> 5) + 3u) >> 2) : ((s.fill >> 7) & ~7u);
    u32x4 *st4 = (u32x4 *)s.stage;
    gptr_u128 dst4 = (gptr_u128)(uintptr_t)(s.out32 + s.base_dw);
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (uint32_t j = lane; j < n4; j += kWave) {
        local_store(st4[j], &dst4[j]); // written once, read much later by another kernel: not worth L2 space (+2 %)
        if (!final) st4[j] = zero4;
    }
    if (!final && n4) {
        // up to 31 complete dwords and the partial one stay: move them to the front of the window
        wave_lds_fence();
        const uint32_t rem = (lane < 32) ? s.stage[4 * n4 + lane] : 0u;
        wave_lds_fence();
        if (lane < 32) s.stage[4 * n4 + lane] = 0u;
        wave_lds_fence();
        if (lane < 32) s.stage[lane] = rem;
        s.base_dw += 4 * n4;
        s.fill -= 128u * n4;
        wave_lds_fence();
    }
}

__device__ __forceinline__ void sink_flush(EmitSink &s, uint32_t lane, bool final)
{
    if (s.wide) {
        sink_flush_exclusive(s, lane, final);
        return;
    }
    wave_lds_fence();
    const uint32_t ndw = final ? ((s.fill + 31) >> 5) : (s.fill >> 5); // (whole 128-byte lines only, as in the 16-byte flush: no gain here)
#pragma unroll 1
    for (uint32_t j = lane; j < ndw; j += kWave) local_store(s.stage[j], &s.out32[s.base_dw + j]);
    if (!final) {
        const uint32_t rem = s.stage[ndw]; // partial dword, uniform address
        wave_lds_fence();
        sink_zero(s, lane, ndw + 3);
        wave_lds_fence();
        if (lane == 0) s.stage[0] = rem;
        s.base_dw += ndw;
        s.fill &= 31;
        wave_lds_fence();
    }
}

// 2-pass histogram in LDS.  Filtered bytes of real images concentrate on a handful of values, and
// LDS atomics from many lanes to ONE address serialise; so the block histogram is replicated 32 times
// with the replica chosen by lane: bank = replica, i.e. every lane owns a bank and at most two lanes
// (l and l+32) ever meet on an address.
constexpr int kHistReplicas = 32;
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t bin, uint32_t lane)
{
    atomicAdd(&hist[bin * kHistReplicas + (lane & (kHistReplicas - 1))], 1u);
}

struct RowResult {
    uint32_t bits;           // token bits of the row
    uint32_t last_unit_bits; // bits of the row's final flush unit (see scan_kernel)
    uint32_t s1, s2;         // Adler raw sums of the filtered row, mod 65521
};

// Walks row r of the job.
template <int C, Pass PASS>
__device__ __forceinline__ RowResult walk_row(const Job &job, const PackedTables &T, uint32_t *hist, uint32_t r, uint32_t lane, EmitSink *sink)
{
    using Raw = typename RowWindows<C>::Raw;
    constexpr int PF = 4; // windows in flight ahead of the one being processed
    constexpr bool kEmit = PASS == Pass::Encode; // builds and stages the token bits
    constexpr bool kSums = PASS == Pass::Encode; // Adler sums, final flush unit
    const uint32_t w = uniform(job.w), bpl = uniform(job.bpl);
    const uint8_t *row = job.rows + (size_t)r * bpl;
    const bool filter_up = (uniform(job.y0) + r) != 0;
    const uint8_t *up_row = filter_up ? (r ? row - bpl : job.row_above) : nullptr;
    const uint32_t filter_byte = filter_up ? 2u : 0u;
    const bool one_pass = uniform(job.one_pass) != 0;
    const bool lit_test = (C == 4) && one_pass; // reference fpng.cpp:1520-1528
    const uint64_t lane_le_mask = (2ull << lane) - 1ull;
    const uint32_t nwin = (w + 63) >> 6;
    const uint32_t n_interior = (w >= 128) ? (w >> 6) - 1 : 0; // windows k with (k+2)*64 <= w

    RowWindows<C> px;
    px.init(row, up_row, bpl, lane);

    Rle<C> rle;
    uint32_t row_bits = 0, last_unit = 0;
    // Adler per-lane accumulators: byte sum, sum of (bytes from the pixel's first byte to the row end) x
    // (pixel byte sum), sum of (byte index inside the pixel) x byte.  s2(row) = acc_w - acc_j.
    uint32_t acc_a = 0, acc_j = 0;
    uint64_t acc_w = 0;
    const uint32_t fl = T.lit[filter_byte]; // filter-type literal in front of pixel 0 (reference fpng.cpp:1473-1475)
    const uint32_t chunk1 = uniform(T.chunk[1]); // token of a 1-pixel chunk (sparse tier)
    if (kEmit) {
        if (lane == 0) {
            const uint64_t v = (uint64_t)plit_code(fl) << (sink->fill & 31);
            atomicOr(&sink->stage[sink->fill >> 5], (uint32_t)v);
            atomicOr(&sink->stage[(sink->fill >> 5) + 1], (uint32_t)(v >> 32));
        }
        sink->fill += plit_len(fl);
    }

    uint32_t f_cur = 0; // state of the per-pixel walk (phase B): filtered pixels and `same` mask of window k
    uint64_t m_cur = 0;

    auto step = [&](auto tail_tag, uint32_t k, uint32_t f_next) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const uint32_t x0 = k << 6;
        const uint32_t last_cur = (uint32_t)__builtin_amdgcn_readlane((int)f_cur, 63);
        uint64_t m_next = __ballot(f_next == lane_prev(f_next, last_cur));
        bool valid = true;
        if (TAIL) {
            m_next &= valid_mask(x0 + 64, w);
            valid = (valid_mask(x0, w) >> lane) & 1;
        }
        uint32_t nbits = 0;
        uint64_t code = 0;
        const bool all_lits = (m_cur == 0); // wave-uniform
        bool every_lane_has_token = all_lits;
        if (all_lits) {
            // no pixel of this window repeats its left neighbour: every lane is a literal pixel
            if (kEmit)
                code = packed_literal_token<C>(T, f_cur, nbits);
            else if (valid) {
                hist_add(hist, f_cur & 0xFF, lane);
                hist_add(hist, (f_cur >> 8) & 0xFF, lane);
                hist_add(hist, (f_cur >> 16) & 0xFF, lane);
                if (C == 4) hist_add(hist, f_cur >> 24, lane);
            }
            rle.carry = 0;
        } else if (!TAIL && rle.carry == 0 && (m_cur & (m_cur >> 1)) == 0 && !((m_cur >> 63) & m_next & 1)) {
            // SPARSE tier: every repeated pixel of this window is an isolated 1-pixel run (no two adjacent
            // mask bits, none continuing into or out of the window).  All lanes build their literal token;
            // the repeated ones swap it for the 1-pixel chunk token (unless the 1-pass RGBA cost rule
            // keeps the literals, reference fpng.cpp:1520-1528).  Typical for photographic content, where
            // the general classification below would otherwise run for one or two lanes' benefit.
            const bool same = (m_cur >> lane) & 1;
            every_lane_has_token = true;
            if (PASS == Pass::Hist) {
                if (same)
                    hist_add(hist, 256 + ((chunk1 >> 8) & 0xFF), lane);
                else {
                    hist_add(hist, f_cur & 0xFF, lane);
                    hist_add(hist, (f_cur >> 8) & 0xFF, lane);
                    hist_add(hist, (f_cur >> 16) & 0xFF, lane);
                    if (C == 4) hist_add(hist, f_cur >> 24, lane);
                }
            } else {
                code = packed_literal_token<C>(T, f_cur, nbits);
                const uint32_t c1_bits = chunk1 & 0xFF;
                if (same && !(lit_test && c1_bits > nbits)) {
                    nbits = c1_bits;
                    code = chunk1 >> 8;
                }
            }
            rle.carry = (uint32_t)(m_cur >> 63); // a run may start on the last lane
        } else if (!TAIL && m_cur == ~0ull) {
            // RUN tier: the whole window lies inside one run.  Chunk boundaries follow from the carry alone.
            uint32_t u = rle.carry + lane; // t - 1
            if (u >= Rle<C>::CAP) u -= Rle<C>::CAP;
            if (C == 4 && u >= Rle<C>::CAP) u -= Rle<C>::CAP;
            const uint32_t q = u + 1;
            const bool ends = (q == Rle<C>::CAP) || (lane == 63 && !(m_next & 1));
            if (ends) {
                if (PASS == Pass::Hist)
                    hist_add(hist, 256 + ((T.chunk[q] >> 8) & 0xFF), lane);
                else {
                    const uint32_t ce = T.chunk[q];
                    nbits = ce & 0xFF;
                    code = ce >> 8;
                    if (lit_test && q == 1) { // a 1-pixel chunk can only be the run's last pixel here
                        uint32_t lbits = 0;
                        const uint64_t lcode = packed_literal_token<C>(T, f_cur, lbits);
                        if (nbits > lbits) {
                            nbits = lbits;
                            code = lcode;
                        }
                    }
                }
            }
            rle.advance(m_cur);
        } else {
            bool ends;
            const uint32_t q = rle.classify(m_cur, (uint32_t)(m_next & 1), lane, lane_le_mask, ends);
            const bool same = (m_cur >> lane) & 1;
            if (PASS == Pass::Hist) {
                if (valid && !same) {
                    hist_add(hist, f_cur & 0xFF, lane);
                    hist_add(hist, (f_cur >> 8) & 0xFF, lane);
                    hist_add(hist, (f_cur >> 16) & 0xFF, lane);
                    if (C == 4) hist_add(hist, f_cur >> 24, lane);
                } else if (ends)
                    hist_add(hist, 256 + ((T.chunk[q] >> 8) & 0xFF), lane); // symbols table: length symbol - 256
            } else {
                uint32_t lbits = 0;
                uint64_t lcode = 0;
                if (!same || (lit_test && ends && q == 1)) lcode = packed_literal_token<C>(T, f_cur, lbits);
                bool as_lits = !same;
                if (ends) {
                    const uint32_t ce = T.chunk[q];
                    nbits = ce & 0xFF;
                    code = ce >> 8;
                    if (lit_test && q == 1 && nbits > lbits) as_lits = true;
                }
                if (as_lits) {
                    nbits = lbits;
                    code = lcode;
                }
            }
            rle.advance(m_cur);
        }
        if (TAIL) {
            if (!valid) {
                nbits = 0;
                code = 0;
            }
            // size of the final flush unit of the row = token of its last pixel (scan_kernel's failure rule)
            if (PASS != Pass::Hist && x0 + 64 >= w) last_unit = (uint32_t)__builtin_amdgcn_readlane((int)nbits, (w - 1) & 63);
        }
        if (kSums) {
            // Adler-32 partial sums (reference fpng.cpp:407-487 computes the same quantity serially).  Lanes
            // past the row end read 0 (RGBA) or are masked (RGB shares an aligned dword with real bytes).
            const uint32_t fa = (!TAIL || C == 4 || valid) ? f_cur : 0u;
            const uint32_t a = __builtin_amdgcn_sad_u8(fa, 0u, 0u);
            acc_a += a;
            acc_w += (uint64_t)(bpl - (uint32_t)C * (x0 + lane)) * a; // invalid lanes: a == 0
            acc_j = __builtin_amdgcn_udot4(fa, 0x03020100u, acc_j, false);
        }
        if (kEmit) {
            const uint32_t incl = wave_inclusive_sum(nbits);
            if (!TAIL && every_lane_has_token)
                sink_put<true>(*sink, code, nbits, sink->fill + incl - nbits);
            else
                sink_put<false>(*sink, code, nbits, sink->fill + incl - nbits);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            sink->fill += total;
            row_bits += total;
            if (sink->fill > (uint32_t)kStageFlushAt * 32u) sink_flush(*sink, lane, false);
        }
        f_cur = f_next;
        m_cur = m_next;
    };

    // =====================================================================================
    // Phase A: 256-pixel super-windows with FOUR consecutive pixels per lane (one buffer_load_dwordx4
    // per lane for the row and one for the Up row; RGB lanes use 12 of the 16 bytes).  The per-window overheads --
    // neighbour compare across lanes, DPP prefix sum, LDS puts, loop -- are paid once per 256 pixels.
    // Only the two cheap tiers are done in this layout (all literal / isolated 1-pixel runs, together
    // ~all of photographic content); any other super-window is replayed through the per-pixel walk
    // below with ds_bpermute gathers.  Super-windows S with 256*(S+1) < w qualify.
    // =====================================================================================
    uint32_t k0 = 0;      // first 64-pixel window left for phase B
    uint32_t carry_f = 0; // filtered value of the pixel just before window k0
    {
        constexpr int ND = C;                         // filtered dwords per lane: 4 pixels x C bytes
        constexpr uint32_t kLaneBytes = 4u * C, kSuperBytes = 256u * C;
        // NS super-windows are followed by at least one more pixel of the row.  A row that ENDS with a complete
        // super-window (w a multiple of 256: 512, 3840, 7680 ...) takes that one here too when it is of a cheap tier: no
        // look-ahead pixel, and the token of the row's last pixel is the final flush unit.  (Taking an INCOMPLETE last
        // super-window here as well -- lanes past the row end masked -- was measured and dropped: 256 x 1080p RGBA 1.15
        // instead of 1.10 ms, and three spilled VGPRs in the 3-channel walk.)
        const uint32_t NS = (w - 1) >> 8;
        const uint32_t NSX = NS + (((w & 255u) == 0) ? 1u : 0u);
        constexpr uint32_t S0 = 0; // the walk's first super-window
        if (NSX > S0) {
            const uint32_t voff4 = lane * kLaneBytes;
            auto load4 = [&](uint32_t S, u32x4 &c4, u32x4 &u4) {
                // RGB: 16 aligned bytes that contain the lane's 12 (the resources start on a dword, the row begins
                // px.phase / px.up_phase bytes into them)
                c4 = __builtin_amdgcn_raw_buffer_load_b128(px.cur, voff4, S * kSuperBytes, 0);
                u4 = __builtin_amdgcn_raw_buffer_load_b128(px.up, voff4, S * kSuperBytes, 0); // (an nt hint on this last use of the row: 0.557 vs 0.500 ms)
            };
            // filtered bytes of the lane's four pixels, packed: fd[0..ND)
            auto filt = [&](const u32x4 &c4, const u32x4 &u4, uint32_t (&fd)[4]) {
                if constexpr (C == 4) {
                    sub_bytes_x4(c4, u4, fd);
                } else {
                    u32x4 ca, ua;
                    ca.x = __builtin_amdgcn_alignbyte(c4.y, c4.x, px.phase);
                    ca.y = __builtin_amdgcn_alignbyte(c4.z, c4.y, px.phase);
                    ca.z = __builtin_amdgcn_alignbyte(c4.w, c4.z, px.phase);
                    ua.x = __builtin_amdgcn_alignbyte(u4.y, u4.x, px.up_phase);
                    ua.y = __builtin_amdgcn_alignbyte(u4.z, u4.y, px.up_phase);
                    ua.z = __builtin_amdgcn_alignbyte(u4.w, u4.z, px.up_phase);
                    ca.w = ua.w = 0;
                    sub_bytes_x4(ca, ua, fd); // (the fourth dword is dead code for the compiler)
                }
            };
            // pixel values (what the per-pixel walk calls f_cur): RGBA = the dwords, RGB = 24-bit fields
            auto pixels = [&](const uint32_t (&fd)[4], uint32_t (&pv)[4]) {
                if constexpr (C == 4) {
                    pv[0] = fd[0], pv[1] = fd[1], pv[2] = fd[2], pv[3] = fd[3];
                } else {
                    pv[0] = fd[0] & 0xFFFFFFu;
                    pv[1] = __builtin_amdgcn_alignbyte(fd[1], fd[0], 3u) & 0xFFFFFFu;
                    pv[2] = __builtin_amdgcn_alignbyte(fd[2], fd[1], 2u) & 0xFFFFFFu;
                    pv[3] = fd[2] >> 8;
                }
            };
            constexpr int PF4 = 1; // super-windows in flight ahead of the look-ahead one
            u32x4 c_first, u_first, rc[PF4], ru[PF4];
            load4(S0, c_first, u_first);
#pragma unroll
            for (int j = 0; j < PF4; j++) load4(S0 + (uint32_t)j + 1, rc[j], ru[j]);
            uint32_t fd[4];
            filt(c_first, u_first, fd);
            // pixel just before the super-window; in front of pixel 0: a value pixel 0 cannot equal (it has no left neighbour)
            uint32_t last_f = ~uniform(fd[0]);
            uint32_t wgt = bpl - kLaneBytes * lane - S0 * kSuperBytes; // bytes from this lane's first byte to the row end
            const uint32_t c1_bits = chunk1 & 0xFF;
            // gather the per-pixel view of 64-pixel window jw of the current super-window (lane i <- pixel 64*jw+i)
            auto gather = [&](uint32_t jw, const uint32_t (&src)[4]) {
                const int sl = (int)((16u * jw + (lane >> 2)) << 2);
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)src[0]);
                const uint32_t t1 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)src[1]);
                const uint32_t t2 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)src[2]);
                const uint32_t t3 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)src[3]);
                const uint32_t comp = lane & 3;
                return comp == 0 ? t0 : (comp == 1 ? t1 : (comp == 2 ? t2 : t3));
            };
            uint32_t gen_streak = 1, done = S0, limit = NSX; // streak starts at 1: a row whose FIRST super-window is general is handed over at once
            for (uint32_t Sb = S0; Sb < limit; Sb += PF4) {
#pragma unroll
                for (int js = 0; js < PF4; js++) {
                    const uint32_t S = Sb + (uint32_t)js;
                    if (S >= limit) break;
                    // look-ahead super-window S+1 (always completely inside the row) out of the ring
                    uint32_t fn[4];
                    filt(rc[js], ru[js], fn);
                    if (S + 1 + PF4 <= NS) load4(S + 1 + PF4, rc[js], ru[js]);
                    uint32_t f[4]; // the four pixels of this lane
                    pixels(fd, f);
                    const uint32_t next_first = (C == 4) ? fn[0] : (fn[0] & 0xFFFFFFu);
                    // per-lane "equals its left neighbour" predicates (their SGPR form is the wave ballot)
                    const bool s0 = f[0] == lane_prev(f[3], last_f); // (pixel 0 of the row: last_f was chosen to differ)
                    const bool s1 = f[1] == f[0], s2 = f[2] == f[1], s3 = f[3] == f[2];
                    const uint64_t M0 = __ballot(s0), M1 = __ballot(s1), M2 = __ballot(s2), M3 = __ballot(s3);
                    const uint32_t last3 = (uint32_t)__builtin_amdgcn_readlane((int)f[3], 63);
                    const bool row_end = S == NS; // (only with w % 256 == 0) the row's last super-window: nothing follows
                    const bool next0 = !row_end && uniform(next_first) == last3; // does the next super-window start by repeating this one's last pixel?
                    const uint64_t any = M0 | M1 | M2 | M3;
                    const bool all_lits = (any == 0);
                    const bool sparse = !all_lits && rle.carry == 0 &&
                                        (((M0 & M1) | (M1 & M2) | (M2 & M3) | (M3 & (M0 >> 1))) == 0) && !((M3 >> 63) && next0);
                    if (all_lits || sparse) {
                        gen_streak = 0;
                        if (PASS == Pass::Hist) {
                            const bool ss[4] = {s0, s1, s2, s3};
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                if (sparse && ss[j])
                                    hist_add(hist, 256 + ((chunk1 >> 8) & 0xFF), lane);
                                else {
                                    hist_add(hist, f[j] & 0xFF, lane);
                                    hist_add(hist, (f[j] >> 8) & 0xFF, lane);
                                    hist_add(hist, (f[j] >> 16) & 0xFF, lane);
                                    if (C == 4) hist_add(hist, f[j] >> 24, lane);
                                }
                            }
                        } else {
                          if (kSums) {
                            // Adler: 4*C consecutive bytes per lane
                            constexpr uint32_t kOffs[4] = {0x03020100u, 0x07060504u, 0x0B0A0908u, 0x0F0E0D0Cu};
                            uint32_t a = 0;
#pragma unroll
                            for (int j = 0; j < ND; j++) {
                                a = __builtin_amdgcn_sad_u8(fd[j], 0u, a);
                                acc_j = __builtin_amdgcn_udot4(fd[j], kOffs[j], acc_j, false);
                            }
                            acc_a += a;
                            acc_w += (uint64_t)wgt * a;
                          }
                          if (kEmit) {
                            // room for a whole super-window (<= 256 x 48 bits = 384 dwords) in the LDS window.  Checked before
                            // the tokens exist: a flush between building them and putting them keeps them all alive across
                            // its loop (spills in the hot path)
                            if (sink->fill > (uint32_t)(kStageDwords - 420) * 32u) sink_flush(*sink, lane, false);
                            uint32_t n[4];
                            uint64_t t[4];
#pragma unroll
                            for (int j = 0; j < 4; j++) t[j] = packed_literal_token<C>(T, f[j], n[j]);
                            if (sparse) {
                                const uint64_t c1_code = chunk1 >> 8;
                                if (M0 && s0 && !(lit_test && c1_bits > n[0])) n[0] = c1_bits, t[0] = c1_code;
                                if (M1 && s1 && !(lit_test && c1_bits > n[1])) n[1] = c1_bits, t[1] = c1_code;
                                if (M2 && s2 && !(lit_test && c1_bits > n[2])) n[2] = c1_bits, t[2] = c1_code;
                                if (M3 && s3 && !(lit_test && c1_bits > n[3])) n[3] = c1_bits, t[3] = c1_code;
                            }
                            if (row_end) last_unit = (uint32_t)__builtin_amdgcn_readlane((int)n[3], 63);
                            const uint32_t nA = n[0] + n[1], nB = n[2] + n[3], nL = nA + nB;
                            const uint32_t incl = wave_inclusive_sum(nL);
                            const uint32_t pos = sink->fill + incl - nL;
                            if (__ballot(nA > 64 || nB > 64)) {
                                // noisy pixels: a pair does not fit 64 bits, put the four tokens one by one
                                sink_put<true>(*sink, t[0], n[0], pos);
                                sink_put<true>(*sink, t[1], n[1], pos + n[0]);
                                sink_put<true>(*sink, t[2], n[2], pos + nA);
                                sink_put<true>(*sink, t[3], n[3], pos + nA + n[2]);
                            } else {
                                sink_put_wide(*sink, t[0] | (t[1] << n[0]), pos);
                                sink_put_wide(*sink, t[2] | (t[3] << n[2]), pos + nA);
                            }
                            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                            sink->fill += total;
                            row_bits += total;
                          }
                        }
                        rle.carry = sparse ? (uint32_t)(M3 >> 63) : 0u;
                    } else {
                        if (row_end) { // the row's last windows need the masked body of phase B
                            limit = S;
                            break;
                        }
                        // general case: replay the super-window as four 64-pixel windows of the per-pixel walk
                        gen_streak++;
                        if (kEmit && sink->fill > (uint32_t)kStageFlushAt * 32u) sink_flush(*sink, lane, false); // room for a 64-pixel window
                        uint32_t fw = gather(0, f);
                        uint32_t carry_px = last_f;
#pragma unroll 1
                        for (uint32_t jw = 0; jw < 4; jw++) {
                            f_cur = fw;
                            m_cur = __ballot(f_cur == lane_prev(f_cur, carry_px));
                            if (S == 0 && jw == 0) m_cur &= ~1ull;
                            carry_px = (uint32_t)__builtin_amdgcn_readlane((int)f_cur, 63);
                            // look-ahead: next window of this super-window, or (only bit 0 is used) the next super-window
                            fw = (jw < 3) ? gather(jw + 1, f) : uniform(next_first);
                            step(std::false_type{}, 4 * S + jw, fw);
                        }
                    }
                    last_f = last3;
                    wgt -= kSuperBytes;
#pragma unroll
                    for (int j = 0; j < 4; j++) fd[j] = fn[j];
                    done = S + 1;
                    // run-length-heavy rows gain nothing from this layout: hand the rest of the row to phase B
                    if (gen_streak >= 2) limit = done;
                }
            }
            k0 = 4 * done;
            carry_f = last_f;
        }
    }

    // =====================================================================================
    // Phase B: per-pixel walk (lane = pixel) of windows k0 .. nwin-1: everything for RGB, the row tail for RGBA
    // =====================================================================================
    if (k0 < nwin) { // (phase A may have taken the whole row)
        if (kEmit && sink->fill > (uint32_t)kStageFlushAt * 32u) sink_flush(*sink, lane, false); // room for a 64-pixel window
        Raw ring[PF];
        const Raw raw0 = px.load_raw(k0 << 6);
#pragma unroll
        for (int j = 0; j < PF; j++) ring[j] = px.load_raw((k0 + (uint32_t)j + 1) << 6);
        f_cur = px.filter(raw0);
        m_cur = __ballot(f_cur == lane_prev(f_cur, carry_f)) & valid_mask(k0 << 6, w);
        if (k0 == 0) m_cur &= ~1ull;
        // interior windows: ring-fed, unmasked
        const uint32_t lim_int = n_interior < nwin ? n_interior : nwin;
        for (uint32_t kb = k0; kb < lim_int; kb += PF) {
#pragma unroll
            for (int j = 0; j < PF; j++) {
                const uint32_t k = kb + (uint32_t)j;
                if (k >= lim_int) break;
                const uint32_t f_next = px.filter(ring[j]);                      // window k+1
                if (k + 1 + PF < nwin) ring[j] = px.load_raw((k + 1 + PF) << 6); // refill the slot
                step(std::false_type{}, k, f_next);
            }
        }
        // the last one or two windows of the row: masked body, look-ahead loaded directly
        for (uint32_t k = (k0 > lim_int ? k0 : lim_int); k < nwin; k++) step(std::true_type{}, k, px.filtered_at((k + 1) << 6));
    }

    RowResult res;
    res.bits = 0;
    res.last_unit_bits = 0;
    res.s1 = res.s2 = 0;
    if (kSums) {
        const uint32_t fl_bits = plit_len(fl);
        res.bits = row_bits + fl_bits;
        // when the row is a single pixel, 1-pass RGB flushes the filter literal together with it
        // (reference fpng.cpp:1186-1203 vs :1473-1497)
        res.last_unit_bits = last_unit + ((C == 3 && one_pass && w == 1) ? fl_bits : 0u);
        const uint32_t la = acc_a % kAdlerMod;
        const uint32_t lw = (uint32_t)((acc_w - acc_j) % kAdlerMod); // every byte weight is positive
        const uint32_t n_mod = (bpl + 1u) % kAdlerMod;
        const uint32_t fb = filter_byte;
        res.s1 = (wave_sum(la) + fb) % kAdlerMod;
        res.s2 = (wave_sum(lw) + n_mod * fb) % kAdlerMod;
    }
    return res;
}

__device__ __forceinline__ const Job &job_of_block(const Job *jobs) { return jobs[blockIdx.y]; }

// hist_kernel (2-pass, pass 1): literal / length-symbol histogram of the whole image
// (reference fpng.cpp:1021-1084 / :1299-1363).  job.table here is the "symbol" table whose
// chunk[q] holds (length symbol - 256).
__device__ __forceinline__ void hist_block(const Job &job, uint32_t *dst)
{
    __shared__ PackedTables T;
    __shared__ uint32_t hist[288 * kHistReplicas];
    if (blockIdx.x * kHistWaves >= job.nrows) return;
    stage_packed_tables<kHistBlock>(T, job.table);
    for (int i = threadIdx.x; i < 288 * kHistReplicas; i += kHistBlock) hist[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, r = blockIdx.x * kHistWaves + uniform(threadIdx.x >> 6);
    if (r < job.nrows) {
        if (job.c == 4)
            walk_row<4, Pass::Hist>(job, T, hist, r, lane, nullptr);
        else
            walk_row<3, Pass::Hist>(job, T, hist, r, lane, nullptr);
        if (lane == 0) hist_add(hist, (job.y0 + r) ? 2 : 0, r); // the row's filter-type literal
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 288; i += kHistBlock) {
        uint32_t s = 0;
        for (int rep = 0; rep < kHistReplicas; rep++) s += hist[i * kHistReplicas + ((rep + i) & (kHistReplicas - 1))];
        if (s) atomicAdd(&dst[i], s);
    }
}
__global__ __launch_bounds__(kHistBlock) __attribute__((amdgpu_num_sgpr(80))) void hist_kernel(const Job *jobs, uint32_t *hist_out)
{
    hist_block(job_of_block(jobs), hist_out + (size_t)blockIdx.y * 288);
}
// (JobArg / the *_first_kernel forms: one image per submission, its job record in the kernel arguments -- see encode_rows_first_kernel)
struct JobArg {
    Job job;
};
__global__ __launch_bounds__(kHistBlock) __attribute__((amdgpu_num_sgpr(80))) void hist_first_kernel(const JobArg arg, Job *job_out, uint32_t *hist_out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *job_out = arg.job; // for build_dynamic_kernel
    hist_block(arg.job, hist_out);
}

// Words that one workgroup writes and another one reads INSIDE a launch: 8-byte granules, agent-scope relaxed atomics
// (write-through stores, L1-bypassing loads; cdna_hip_programming.md Guideline 16, R2).
typedef FPNG_GLOBAL uint64_t *gptr_u64;
#define FPNG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ uint64_t granule_load(const uint64_t *p)
{
    return __hip_atomic_load((gptr_u64)(uintptr_t)p, FPNG_RLX_AGENT);
}
__device__ __forceinline__ void granule_store(uint64_t *p, uint64_t v)
{
    __hip_atomic_store((gptr_u64)(uintptr_t)p, v, FPNG_RLX_AGENT);
}

// ---------------------------------------------------------------------------------------------
// Row scan of one job by one 256-thread block: exclusive scan of the rows' token bits -> absolute bit offset of every
// row, Adler combine, the reference's compressed-or-stored decision, sizes, head of the output.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_exclusive_sum_u64(uint64_t v, uint32_t lane, uint64_t &total)
{
    uint64_t incl = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint64_t up = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), o, kWave) << 32) | (uint32_t)__shfl_up((int)(uint32_t)incl, o, kWave);
        if ((int)lane >= o) incl += up;
    }
    total = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(incl >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)incl, 63);
    return incl - v;
}

template <typename P> __device__ __forceinline__ void store_be32(P p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

__device__ __forceinline__ void scan_job(const Job &job, JobState &st, const RowInfo *rows, uint64_t *row_off, uint32_t n_jobs,
                                         uint64_t (*wsum)[kScanWaves] /* LDS [3][kScanWaves] */)
{
    const uint32_t n_rec = job.nrows;
    const uint32_t t = threadIdx.x, lane = t & 63, wv = uniform(t >> 6);
    const TokenTable *tab = job.table;
    const uint64_t n_filtered = (uint64_t)(job.bpl + 1) * job.nrows;
    const bool force_stored = (job.flags & 2u) != 0;
    const FPNG_GLOBAL u32x4 *rg = (const FPNG_GLOBAL u32x4 *)(uintptr_t)(rows + job.row_base); // {bits, s1, s2, -} per row

    // --- exclusive scan of row bits (absolute zlib bit positions), Adler combine ---
    const uint64_t first_bit = job.is_first ? tab->first_token_bit : job.start_bit;
    uint64_t s_last = first_bit; // zlib bit position after the last token
    uint64_t a_s1 = 0, a_s2 = 0;
    const uint32_t n_row_mod = (job.bpl + 1u) % kAdlerMod;
    if (!force_stored) {
        // every thread owns a contiguous chunk of rows: local sums, one scan of the 256 chunk totals (per wave, then across
        // the four waves through LDS), then the chunk is walked again to hand out the row offsets
        const uint32_t per = (n_rec + kScanBlock - 1) / kScanBlock;
        const uint32_t r0 = t * per < n_rec ? t * per : n_rec, r1 = (r0 + per < n_rec) ? r0 + per : n_rec;
        uint64_t local = 0;
        for (uint32_t rb = r0; rb < r1; rb += 8) { // eight records in flight per round trip
            u32x4 ri[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) ri[k] = rg[rb + k < r1 ? rb + k : r1 - 1];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t r = rb + k;
                if (r >= r1) break;
                const uint32_t s1 = ri[k].y, s2 = ri[k].z; // < 65521
                local += ri[k].x;
                // S2 of the concatenation: every byte of this row is followed by the later rows.  All factors are below
                // 65521, so the products (and 65520^2 + 65520) fit 32 bits
                const uint32_t after = ((job.nrows - 1 - r) % kAdlerMod) * n_row_mod % kAdlerMod;
                a_s1 += s1;
                a_s2 += (s2 + after * s1) % kAdlerMod;
            }
        }
        uint64_t wave_total;
        const uint64_t excl = wave_exclusive_sum_u64(local, lane, wave_total);
        if (lane == 0) wsum[0][wv] = wave_total;
        __syncthreads();
        uint64_t before = 0, total = 0;
#pragma unroll
        for (uint32_t k = 0; k < kScanWaves; k++) {
            const uint64_t v = wsum[0][k];
            if (k < wv) before += v;
            total += v;
        }
        uint64_t pos = first_bit + before + excl;
        for (uint32_t rb = r0; rb < r1; rb += 4) {
            uint32_t bits[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) bits[k] = rg[rb + k < r1 ? rb + k : r1 - 1].x;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                if (rb + k >= r1) break;
                row_off[job.row_base + rb + k] = pos;
                pos += bits[k];
            }
        }
        s_last += total;
    }
    {
        const uint32_t w1 = wave_sum((uint32_t)(a_s1 % kAdlerMod)), w2 = wave_sum((uint32_t)(a_s2 % kAdlerMod));
        if (lane == 0) wsum[1][wv] = w1, wsum[2][wv] = w2;
    }
    __syncthreads();
    uint64_t S1 = 0, S2 = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanWaves; k++) S1 += wsum[1][k], S2 += wsum[2][k];
    S1 %= kAdlerMod, S2 %= kAdlerMod;

    // --- compressed or stored?  (closed form of reference fpng.cpp:567-588) ---
    const uint32_t eob_len = tab->lit[256] >> 16;
    // byte budget the reference hands to the coder (fpng.cpp:1705): whole image only
    const uint64_t n_total = (uint64_t)(job.bpl + 1) * job.h_total;
    const uint64_t D = ((58 + n_total + 7) & ~7ull) - 58;
    const uint32_t last_unit_bits = st.last_unit_bits;
    bool stored = force_stored;
    if (job.whole_png && !force_stored) {
        if (job.one_pass && D < tab->header_bits / 8) stored = true;                 // fpng.cpp:1169, :1455
        if (((s_last - last_unit_bits) >> 3) + 8 > D) stored = true;                  // last PUT_BITS_FLUSH
        if (((s_last + eob_len + 7) >> 3) + 4 > D) stored = true;                    // EOB + Adler
    }
    const uint64_t zlib_bytes_no_adler = stored ? (2 + n_filtered + 5 * ((n_filtered + kStoredBlockMax - 1) / kStoredBlockMax))
                                                : ((s_last + eob_len + 7) >> 3);
    const uint64_t zlib_size = (!job.whole_png && job.band_zlib_size) ? job.band_zlib_size : zlib_bytes_no_adler + 4;

    if (t == 0) {
        st.token_end_bit = s_last;
        st.mode = stored ? 1u : 0u;
        st.status = 0;
        st.zlib_size = zlib_size;
        st.s1 = (uint32_t)S1;
        st.s2 = (uint32_t)S2;
        const uint32_t a1 = (uint32_t)((1 + S1) % kAdlerMod);
        const uint32_t a2 = (uint32_t)((n_filtered % kAdlerMod + S2) % kAdlerMod);
        st.adler = (a2 << 16) | a1; // Adler-32 of the Up/None-filtered stream (compressed mode)
        // bytes per assemble/crc block: 64 KiB when the submission has plenty of blocks anyway (every block loads
        // the 16 KiB CRC table), down to one 4 KiB block row for small files so that the submission still spreads
        // over ~2048 blocks
        const uint64_t span = kPngHeaderBytes + zlib_size; // >= aligned data end - 48
        uint32_t want = 2048u / n_jobs;                    // blocks this job should get (512 / 256 / 128 measured: no better for single frames)
        want = want < 4u ? 4u : want;
        uint32_t rl = 12;
        while (rl < 16 && (((span >> rl) + 1 > want) || ((span >> rl) + 1 > job.crc_blocks))) rl++;
        if (!job.whole_png) rl = 16; // row bands: every rank must cut the file into the same ranges (their CRC partials are XOR-ed)
        st.range_log2 = rl;
    }
    // a band's counting phase stops here; whole images and band placements prepare the head of the output
    if (!job.whole_png && !(job.flags & 0x100u)) return;

    // --- PNG header + Deflate block header; assemble_kernel, which places the rows, wants the head followed by zeros
    //     up to the next 16-byte boundary ---
    gptr_u8 out = to_global<gptr_u8>(job.out);
    if (job.whole_png)
        for (uint32_t i = t; i < kPngHeaderBytes; i += kScanBlock)
            if (i < 50 || i >= 54) out[i] = job.png_header[i];
    gptr_u8 zl = out + (job.bit_bias >> 3); // zlib byte 0 (only meaningful for the first band / whole image)
    if (!stored && job.is_first) {
        const uint32_t head_bytes = (tab->header_bits + 7) >> 3; // (the last one holds the pending bits in front of the first token)
        for (uint32_t i = t; i < head_bytes; i += kScanBlock) zl[i] = tab->header[i];
        const uint32_t head_end = kPngHeaderBytes + head_bytes;
        for (uint32_t i = head_end + t; i < ((head_end + 15u) & ~15u); i += kScanBlock) out[i] = 0;
    }
    if (t == 0 && job.whole_png) store_be32(out + 50, (uint32_t)zlib_size); // IDAT length (reference fpng.cpp:1782)
}

// scan_kernel: one block per job (whole images; row bands: counting phase and placement phase)
__global__ __launch_bounds__(kScanBlock) void scan_kernel(const Job *jobs, const RowInfo *rows, uint64_t *row_off, JobState *states)
{
    __shared__ uint64_t wsum[3][kScanWaves];
    scan_job(jobs[blockIdx.x], states[blockIdx.x], rows, row_off, gridDim.x, wsum);
}

// ---------------------------------------------------------------------------------------------
// encode_rows_kernel: grid (ceil(max_rows/8), n_jobs).  ONE walk over the pixels of a whole image: each wave
// encodes its row into the row's private, dword-aligned local stream (plain coalesced stores, nothing is
// shared between rows) and records the row's token bits and Adler sums for scan_kernel.
// ---------------------------------------------------------------------------------------------
// One instantiation per channel count (jobs of the other kind leave at once): the 3-channel walk needs far
// fewer registers than the 4-pixels-per-lane RGBA one and keeps 8 waves per SIMD.
template <int C>
__device__ __forceinline__ void encode_rows_block(const Job &job, uint32_t by, uint32_t bx, RowInfo *rows_out, JobState *states, uint32_t *local)
{
    __shared__ PackedTables T;
    __shared__ __attribute__((aligned(16))) uint32_t stage[kRowWaves][kStageDwords + 2 * kWave + 4]; // + dump slots, see sink_put
    if (job.c != C || bx * kRowWaves >= job.nrows) return;
    stage_packed_tables<kRowBlock>(T, job.table);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wv = uniform(threadIdx.x >> 6), r = bx * kRowWaves + wv;
    if (r >= job.nrows) return;

    EmitSink sink;
    sink.stage = stage[wv];
    sink.out32 = to_global<gptr_u32>(local + job.local_base + (uint64_t)r * job.local_stride);
    // (a zero the compiler cannot see: with a literal 0 it specialises the walk on the known fill level and
    // nearly doubles the register count)
    const uint32_t zero = uniform(job.local_pad);
    sink.base_dw = zero;
    sink.fill = zero;
    sink.wide = (C == 4); // the 3-channel walk has no registers to spare for the 16-byte flush (it would drop to 7 waves/SIMD)
    if (C == 3)
        sink_zero_window(sink, lane); // (in the 4-channel kernel the wider stores cost two more spilled registers in the walk)
    else
        sink_zero(sink, lane, kStageDwords);
    wave_lds_fence();

    const RowResult res = walk_row<C, Pass::Encode>(job, T, nullptr, r, lane, &sink);
    if (r == job.nrows - 1 && job.is_last) {
        // end of block symbol behind the last row's tokens (reference fpng.cpp:1564-1567); not part of ri.bits
        const uint32_t eob = T.lit[256];
        sink_put<false>(sink, lane == 0 ? (uint64_t)plit_code(eob) : 0ull, lane == 0 ? plit_len(eob) : 0u, sink.fill);
        sink.fill += plit_len(eob);
    }
    sink_flush(sink, lane, true);
    if (lane == 0) {
        RowInfo ri;
        ri.bits = res.bits;
        ri.s1 = res.s1;
        ri.s2 = res.s2;
        ri.pad = 0;
        rows_out[job.row_base + r] = ri;
        if (r == job.nrows - 1) states[by].last_unit_bits = res.last_unit_bits;
    }
}

// XCD-aware order: workgroups go round-robin to the 8 XCDs (each with its own L2).  Hand every XCD a
// contiguous range of (job, row block) pairs, so that the block holding the row above a block's first row runs
// on the same XCD at about the same time and that row is an L2 hit rather than a second HBM read.
__device__ __forceinline__ void xcd_block_order(uint32_t &bx, uint32_t &by)
{
    const uint32_t total = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t xcd = lin & 7u, per = total >> 3, rem = total & 7u;
    const uint32_t logical = xcd * per + (xcd < rem ? xcd : rem) + (lin >> 3);
    by = logical / gridDim.x;
    bx = logical - by * gridDim.x;
}

template <int C, int WPE>
__global__ __launch_bounds__(kRowBlock) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(WPE, WPE))) void encode_rows_kernel(const Job *jobs, RowInfo *rows_out,
                                                                                                      JobState *states, uint32_t *local)
{
    uint32_t bx, by;
    xcd_block_order(bx, by);
    encode_rows_block<C>(jobs[by], by, bx, rows_out, states, local);
}

// One image per submission, first kernel of its chain: the job record comes IN THE KERNEL ARGUMENTS instead of through an
// upload in front of the chain (a blit kernel + a dispatch gap: ~7 us of a single frame's ~90); workgroup 0 leaves it in
// device memory for scan / assemble / finalize, which start after this kernel has ended.
template <int C, int WPE>
__global__ __launch_bounds__(kRowBlock) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(WPE, WPE))) void encode_rows_first_kernel(const JobArg arg, Job *job_out,
                                                                                                            RowInfo *rows_out, JobState *states,
                                                                                                            uint32_t *local)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *job_out = arg.job; // (one lane, constant offsets: indexing the argument by thread would put a copy of it into scratch)
    uint32_t bx, by;
    xcd_block_order(bx, by);
    encode_rows_block<C>(arg.job, 0, bx, rows_out, states, local);
}

// ---------------------------------------------------------------------------------------------
// crc_kernel: raw (init 0, no final xor) CRC-32 partials of the zlib bytes except the 4 Adler
// bytes (reference fpng.cpp:234-292 computes the same function serially / with pclmul).
//
// grid (max_crc_blocks, n_jobs).  Block j owns the 64 KiB range that ENDS j*64 KiB before the
// 16-byte-aligned end of the data, so every lane does aligned 16-byte loads and the only padding
// is < 16 zero bytes at the very end (undone with one constant in finalize_kernel); bytes in front
// of the data count as zero, which a raw CRC ignores.  Per step the 256 lanes cover one 4 KiB block
// row; each lane keeps the CRC of its own 16-byte stripe with slice-by-16 tables that already
// contain the 4080-byte jump to its next piece.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t crc_range_log2(const JobState &st) { return st.range_log2 ? st.range_log2 : 16u; }

__device__ __forceinline__ uint32_t dev_mulmod(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll 8
    for (int i = 31; i >= 0; i--) {
        r ^= b & (0u - ((a >> i) & 1u));
        b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
    }
    return r;
}

__global__ __launch_bounds__(kBlock) void crc_kernel(const Job *jobs, const JobState *states, const CrcDeviceTables *tabs,
                                                    uint32_t *partials, uint32_t max_crc_blocks)
{
    __shared__ uint32_t tab[16][256];
    __shared__ uint32_t red[kWavesPerBlock];
    const Job &job = job_of_block(jobs);
    const JobState &st = states[blockIdx.y];
    if (!job.whole_png) return;
    const int64_t data_begin = kPngHeaderBytes, data_end = (int64_t)(kPngHeaderBytes + st.zlib_size - 4);
    const int64_t end_aligned = (data_end + 15) & ~15ll;
    const uint32_t range_bytes = 1u << crc_range_log2(st);
    const int64_t range_end = end_aligned - (int64_t)blockIdx.x * range_bytes;
    if (range_end <= (data_begin & ~15ll)) return; // nothing of the data in this range
    for (int i = threadIdx.x; i < 16 * 256; i += kBlock) (&tab[0][0])[i] = (&tabs->striped[0][0])[i];
    __syncthreads();
    gptr_cu8 base = to_global<gptr_cu8>(job.out);
    const uint32_t tid = threadIdx.x;
    uint32_t c = 0;
    for (uint32_t row = 0; row < range_bytes / kCrcRowBytes; row++) {
        const int64_t o = range_end - range_bytes + (int64_t)row * kCrcRowBytes + tid * 16;
        uint32_t w[4] = {0, 0, 0, 0};
        if (o + 16 > data_begin && o < data_end) {
            const u32x4 d = *(gptr_cu128)(base + o);
            w[0] = d.x, w[1] = d.y, w[2] = d.z, w[3] = d.w;
            if (o < data_begin || o + 16 > data_end) { // zero the bytes outside [data_begin, data_end)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t m = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int64_t pos = o + 4 * k + b;
                        if (pos >= data_begin && pos < data_end) m |= 0xFFu << (8 * b);
                    }
                    w[k] &= m;
                }
            }
        }
        w[0] ^= c;
        uint32_t n = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            n ^= tab[4 * k + 0][w[k] & 0xFF] ^ tab[4 * k + 1][(w[k] >> 8) & 0xFF] ^ tab[4 * k + 2][(w[k] >> 16) & 0xFF] ^
                 tab[4 * k + 3][w[k] >> 24];
        c = n;
    }
    // lane stripes now sit at range_end + 16*tid: move them all to range_end + one block row, fold
    c = wave_xor(dev_mulmod(c, tabs->lane_fix[tid]));
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) partials[(size_t)blockIdx.y * max_crc_blocks + blockIdx.x] = red[0] ^ red[1] ^ red[2] ^ red[3];
}

// ---------------------------------------------------------------------------------------------
// finalize_kernel: one block per job
// ---------------------------------------------------------------------------------------------
// four independent products, interleaved (the single product is a chain of 32 dependent steps)
__device__ __forceinline__ void dev_mulmod4(const uint32_t (&a)[4], const uint32_t (&b_in)[4], uint32_t (&r)[4])
{
    uint32_t b[4] = {b_in[0], b_in[1], b_in[2], b_in[3]};
    r[0] = r[1] = r[2] = r[3] = 0;
#pragma unroll 4
    for (int i = 31; i >= 0; i--) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r[k] ^= b[k] & (0u - ((a[k] >> i) & 1u));
            b[k] = (b[k] >> 1) ^ (0xEDB88320u & (0u - (b[k] & 1u)));
        }
    }
}
__device__ __forceinline__ uint32_t dev_crc_byte(uint32_t c, uint32_t byte)
{
    c ^= byte;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    return c;
}

__device__ __forceinline__ uint64_t block_sum_u64(uint64_t v, uint64_t *red64)
{
    const uint32_t t = threadIdx.x;
    red64[t] = v;
    __syncthreads();
    for (uint32_t o = kBlock / 2; o > 0; o >>= 1) {
        if (t < o) red64[t] += red64[t + o];
        __syncthreads();
    }
    const uint64_t r = red64[0];
    __syncthreads();
    return r;
}

// One block (kBlock threads) finishes one image: CRC fold, Adler, IDAT CRC, IEND, result record.  `partials` of a
// launch that calls this from its own last block were written by other workgroups: granule-style loads.
__device__ __forceinline__ void finalize_job(const Job &job, const RowInfo *rows, JobState &st, const CrcDeviceTables *tabs,
                                             const uint32_t *pj, const uint32_t *aj, Result &result, uint32_t *red, uint64_t *red64)
{
    const uint32_t t = threadIdx.x;
    if (!job.whole_png) {
        if (t == 0) {
            result.png_size = 0;
            result.mode = st.mode;
            result.status = 0;
        }
        return;
    }
    const uint64_t zlib_size = st.zlib_size;
    const int64_t data_end = (int64_t)(kPngHeaderBytes + zlib_size - 4);
    const int64_t end_aligned = (data_end + 15) & ~15ll;
    const uint32_t rl = crc_range_log2(st);
    const uint32_t n_ranges = (uint32_t)((end_aligned - 48 + (1ll << rl) - 1) >> rl);
    uint32_t adler = st.adler;
    if (st.mode == 1u && aj) {
        // stored mode: Adler-32 of the filter-0 stream from the ranges' sums (assemble_stored): byte sums, and sums weighted with
        // the bytes from each byte to the end of the stream
        uint64_t s1 = 0, s2 = 0;
        for (uint32_t j = t; j < n_ranges; j += kBlock) s1 += aj[2 * j], s2 += aj[2 * j + 1];
        const uint64_t S1 = block_sum_u64(s1 % kAdlerMod, red64) % kAdlerMod;
        const uint64_t S2 = block_sum_u64(s2 % kAdlerMod, red64) % kAdlerMod;
        const uint64_t n_filtered = (uint64_t)(job.bpl + 1) * job.nrows;
        adler = (uint32_t)(((n_filtered % kAdlerMod + S2) % kAdlerMod) << 16) | (uint32_t)((1 + S1) % kAdlerMod);
    }

    // ---- fold the CRC partials.  Partial j sits (j ranges + one block row) before the common end
    //      point, so  T = XOR_j p_j * X^j  with X = x^(8*64Ki); all needed constants are x^(8*2^i). ----
    uint32_t g = 0; // each thread folds G = 2^g consecutive partials
    while (((uint64_t)kBlock << g) < n_ranges) g++;
    const uint32_t G = 1u << g;
    // (the constants of the later steps are asked for now: their loads travel together with those of the partials instead of
    // one round trip each behind the fold)
    const uint32_t group_pow = tabs->fold[rl + g - 12][t];
    const uint32_t len_pow = (t < 6) ? tabs->pow_byte[t][((zlib_size - 4) >> (8 * t)) & 0xFF] : 0x80000000u; // 0x80000000 = 1
    const uint32_t unpad = tabs->inv_row_pad[(uint32_t)(end_aligned - data_end)];
    uint32_t v = 0;
    {
        // partial i of the group times x^(8*range*i): independent multiplications, four at a time (a Horner chain would
        // be G dependent ones: G = 32 for a 16384^2 image)
        const uint32_t *xp = tabs->fold[rl - 12];
        for (uint32_t i = 0; i < G; i += 4) {
            uint32_t a[4], b[4], r[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t j = t * G + i + k;
                a[k] = (i + k < G && j < n_ranges) ? pj[j] : 0u;
                b[k] = xp[(i + k) & 255u];
            }
            dev_mulmod4(a, b, r);
            v ^= r[0] ^ r[1] ^ r[2] ^ r[3];
        }
    }
    // the thread's group starts t * G ranges before the common end point: one multiplication by a tabulated power, then
    // the groups simply XOR together (no multiplications inside the reduction)
    if (v && t) v = dev_mulmod(v, group_pow);
    v = wave_xor(v);
    // x^(8*(zlib_size-4)): six tabulated factors (one per byte of the length), multiplied as a tree by lanes 0..7 of wave 0
    uint32_t f = len_pow;
    if (t < 64) {
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_down((int)f, o, 8);
            f = dev_mulmod(f, other);
        }
    }
    if ((t & 63) == 0) red[t >> 6] = v;
    if (t == 0) red[kWavesPerBlock] = f;
    __syncthreads();
    const uint32_t folded = red[0] ^ red[1] ^ red[2] ^ red[3];
    if (t == 0) {
        gptr_u8 out = to_global<gptr_u8>(job.out);
        const uint32_t raw_data = dev_mulmod(folded, unpad);
        // running CRC state (init ~0) after "IDAT", advanced over the data, then the 4 Adler bytes
        uint32_t s = 0xFFFFFFFFu;
        s = dev_crc_byte(s, 'I');
        s = dev_crc_byte(s, 'D');
        s = dev_crc_byte(s, 'A');
        s = dev_crc_byte(s, 'T');
        s = dev_mulmod(s, red[kWavesPerBlock]) ^ raw_data;
        gptr_u8 tail = out + kPngHeaderBytes + zlib_size - 4;
        store_be32(tail, adler); // reference fpng.cpp:1569-1577 / :851-863
        s = dev_crc_byte(s, adler >> 24);
        s = dev_crc_byte(s, (adler >> 16) & 0xFF);
        s = dev_crc_byte(s, (adler >> 8) & 0xFF);
        s = dev_crc_byte(s, adler & 0xFF);
        const uint32_t crc = ~s;
        store_be32(tail + 4, crc); // reference fpng.cpp:1797-1800
        const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
        for (int i = 0; i < 12; i++) tail[8 + i] = iend[i];
        st.adler = adler;
        st.crc = crc;
        result.png_size = kPngHeaderBytes + zlib_size + kPngTrailerBytes;
        result.mode = st.mode;
        result.status = st.status;
    }
}

__global__ __launch_bounds__(kBlock) void finalize_kernel(const Job *jobs, const RowInfo *rows, JobState *states,
                                                         const CrcDeviceTables *tabs, const uint32_t *partials, const uint32_t *adler_parts,
                                                         uint32_t max_crc_blocks, Result *results)
{
    __shared__ uint32_t red[kBlock];
    __shared__ uint64_t red64[kBlock];
    finalize_job(jobs[blockIdx.x], rows, states[blockIdx.x], tabs, partials + (size_t)blockIdx.x * max_crc_blocks,
                 adler_parts ? adler_parts + 2 * (size_t)blockIdx.x * max_crc_blocks : nullptr, results[blockIdx.x], red, red64);
}

// ---------------------------------------------------------------------------------------------
// assemble_kernel: same geometry and CRC arithmetic as crc_kernel, but the 16 bytes a lane feeds to its CRC
// stripe are ASSEMBLED here from the rows' local streams and stored to the file: row r's stream is shifted
// to file bit row_off[r] + bias, neighbouring rows meet inside a dword.  Nothing is read back from the file
// except the head (PNG header + Deflate prefix, written by scan_kernel), and no destination dword is
// written twice, so the rows need no atomics and no zeroed seams.  Stored-mode jobs only take the CRC.
// ---------------------------------------------------------------------------------------------

// Stored-block fallback (reference fpng.cpp:818-866 over the filter-0 stream, :1728-1758), done by the same workgroups that
// would otherwise place the compressed rows: the block's range of the FILE is produced straight from the pixels --
//   zlib offset z:  0,1 = 78 01;  stored block k: header at 2 + 65540 k (BFINAL, LEN, ~LEN), data = stream bytes
//   [65535 k, ...) behind it;  stream byte s = filter byte 0 (s % (bpl+1) == 0) or pixel byte (row s / (bpl+1), s % (bpl+1) - 1)
// -- together with its CRC partial and its share of the Adler-32 (byte sum, position-weighted sum: finalize_kernel adds the
// ranges up).  A piece of 16 file bytes that lies inside one stored block and one row is 16 consecutive pixel bytes (5 aligned
// loads + v_alignbyte); pieces with a block header, a filter byte or the ends of the stream are built byte by byte.
__device__ __forceinline__ uint32_t stored_stream_byte(const Job &job, gptr_cu8 px, uint32_t s)
{
    const uint32_t stride = job.bpl + 1, row = s / stride, col = s - row * stride;
    return col ? (uint32_t)px[(size_t)row * job.bpl + col - 1] : 0u;
}

__device__ __forceinline__ void assemble_stored(const Job &job, const JobState &st, int64_t range_begin, uint32_t range_bytes, int32_t db, int32_t de,
                                                uint32_t (*tab)[256], uint32_t *red, const CrcDeviceTables *tabs, uint32_t *crc_out, uint32_t *adler_out)
{
    const uint32_t tid = threadIdx.x, stride = job.bpl + 1;
    const uint32_t n_filtered = stride * job.nrows; // (< 2^32: check_dims)
    gptr_cu8 px = to_global<gptr_cu8>(job.rows);
    gptr_u8 file = to_global<gptr_u8>(job.out);
    // the lane's position in the stored-block structure, kept up from step to step (one step = 4096 file bytes further)
    int64_t z = range_begin + (int64_t)tid * 16 - kPngHeaderBytes; // zlib offset of the lane's piece
    uint64_t k = 0;
    uint32_t w = 0; // z = 2 + 65540 k + w  (valid when z >= 2)
    if (z >= 2) k = (uint64_t)(z - 2) / 65540u, w = (uint32_t)((uint64_t)(z - 2) - k * 65540u);
    uint32_t c = 0;
    uint64_t a_sum = 0, w_sum = 0;
    for (uint32_t row = 0; row < range_bytes / kCrcRowBytes; row++) {
        const int32_t o = (int32_t)(row * kCrcRowBytes + tid * 16);
        uint32_t v[4] = {0, 0, 0, 0};
        if (o + 16 > db && o < de) {
            const int64_t fo = range_begin + o;
            bool fast = false;
            if (z >= 7 && w >= 5 && w <= 65540u - 16u) {
                const uint64_t s0 = k * 65535u + (w - 5);
                if (s0 + 16 <= n_filtered) {
                    const uint32_t s32 = (uint32_t)s0, r = s32 / stride, col = s32 - r * stride;
                    if (col >= 1 && col + 16 <= stride) { // 16 consecutive pixel bytes
                        fast = true;
                        gptr_cu8 p = px + (size_t)r * job.bpl + (col - 1);
                        const uint32_t m = (uint32_t)(uintptr_t)p & 3u;
                        gptr_cu32 q = (gptr_cu32)(uintptr_t)(p - m);
                        const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = m ? q[4] : 0u; // (never outside the image's bytes' words)
                        v[0] = __builtin_amdgcn_alignbyte(q1, q0, m), v[1] = __builtin_amdgcn_alignbyte(q2, q1, m);
                        v[2] = __builtin_amdgcn_alignbyte(q3, q2, m), v[3] = __builtin_amdgcn_alignbyte(q4, q3, m);
                        uint32_t a = 0, dj = 0;
                        constexpr uint32_t kOffs[4] = {0x03020100u, 0x07060504u, 0x0B0A0908u, 0x0F0E0D0Cu};
#pragma unroll
                        for (int t = 0; t < 4; t++) a = __builtin_amdgcn_sad_u8(v[t], 0u, a), dj = __builtin_amdgcn_udot4(v[t], kOffs[t], dj, false);
                        a_sum += a;
                        w_sum += (uint64_t)(n_filtered - s32) * a - dj; // weight of stream byte s: bytes from it to the end of the stream
                    }
                }
            }
            if (!fast) {
                uint64_t kk = k;
                uint32_t ww = w;
                for (int j = 0; j < 16; j++) {
                    const int64_t zj = z + j;
                    uint32_t b = 0;
                    if (zj == 0)
                        b = 0x78;
                    else if (zj == 1)
                        b = 0x01;
                    else if (zj >= 2) {
                        if (zj == 2) kk = 0, ww = 0;
                        if (ww < 5) { // header of stored block kk
                            const uint64_t done = kk * 65535u;
                            if (done < n_filtered) {
                                const uint32_t remaining = n_filtered - (uint32_t)done, len = remaining < kStoredBlockMax ? remaining : kStoredBlockMax;
                                b = ww == 0 ? (remaining <= kStoredBlockMax ? 1u : 0u) : ww == 1 ? (len & 0xFF) : ww == 2 ? (len >> 8) : ww == 3 ? (~len & 0xFF) : ((~len >> 8) & 0xFF);
                            }
                        } else {
                            const uint64_t sj = kk * 65535u + (ww - 5);
                            if (sj < n_filtered) {
                                b = stored_stream_byte(job, px, (uint32_t)sj);
                                a_sum += b;
                                w_sum += (uint64_t)(n_filtered - (uint32_t)sj) * b;
                            }
                        }
                        if (++ww == 65540u) ww = 0, kk++;
                    }
                    v[j >> 2] |= b << (8 * (j & 3));
                }
            }
            if (fo >= 64) { // (a whole piece behind the PNG header; bytes behind the data are finalize_kernel's, written afterwards)
                u32x4 d;
                d.x = v[0], d.y = v[1], d.z = v[2], d.w = v[3];
                *(gptr_u128)(uintptr_t)(file + fo) = d;
            } else {
                for (int j = 0; j < 16; j++)
                    if (fo + j >= (int64_t)kPngHeaderBytes) file[fo + j] = (uint8_t)(v[j >> 2] >> (8 * (j & 3)));
            }
            if (o < db || o + 16 > de) { // the CRC covers [data_begin, data_end)
#pragma unroll
                for (int kq = 0; kq < 4; kq++) {
                    uint32_t msk = 0;
#pragma unroll
                    for (int bq = 0; bq < 4; bq++) {
                        const int32_t pos = o + 4 * kq + bq;
                        if (pos >= db && pos < de) msk |= 0xFFu << (8 * bq);
                    }
                    v[kq] &= msk;
                }
            }
        }
        v[0] ^= c;
        uint32_t nx = 0;
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
            nx ^= tab[4 * kq + 0][v[kq] & 0xFF] ^ tab[4 * kq + 1][(v[kq] >> 8) & 0xFF] ^ tab[4 * kq + 2][(v[kq] >> 16) & 0xFF] ^ tab[4 * kq + 3][v[kq] >> 24];
        c = nx;
        // one step further in the file
        z += kCrcRowBytes;
        if (z >= 2) {
            if (z - (int64_t)kCrcRowBytes < 2)
                k = (uint64_t)(z - 2) / 65540u, w = (uint32_t)((uint64_t)(z - 2) - k * 65540u);
            else if ((w += kCrcRowBytes) >= 65540u)
                w -= 65540u, k++;
        }
    }
    c = wave_xor(dev_mulmod(c, tabs->lane_fix[tid]));
    const uint32_t s1 = wave_sum((uint32_t)(a_sum % kAdlerMod)), s2 = wave_sum((uint32_t)(w_sum % kAdlerMod));
    if ((tid & 63) == 0) red[tid >> 6] = c, red[kWavesPerBlock + (tid >> 6)] = s1, red[2 * kWavesPerBlock + (tid >> 6)] = s2;
    __syncthreads();
    if (tid == 0) {
        uint32_t x = 0, t1 = 0, t2 = 0;
        for (int q = 0; q < kWavesPerBlock; q++) x ^= red[q], t1 += red[kWavesPerBlock + q], t2 += red[2 * kWavesPerBlock + q];
        *crc_out = x;
        adler_out[0] = t1 % kAdlerMod, adler_out[1] = t2 % kAdlerMod;
    }
}

// (measured and dropped, round 5: assemble_kernel capped at 4 / 2 waves per SIMD so that the other lane's row walk finds room next to it --
//  8 x 8K 0.49 -> 0.54 / 0.69 ms per step, profiles/r05_rows_w6.txt block 5)
__global__ __launch_bounds__(kBlock) void assemble_kernel(const Job *jobs, JobState *states, const uint64_t *row_off,
                                                         const uint32_t *local, const CrcDeviceTables *tabs, uint32_t *partials,
                                                         uint32_t *adler_parts, uint32_t max_crc_blocks)
{
    __shared__ uint32_t tab[16][256];
    __shared__ uint32_t red[3 * kWavesPerBlock];
    const Job &job = job_of_block(jobs);
    JobState &st = states[blockIdx.y];
    // row bands (flag 0x100): the rows of a band land in a private window that shares the whole file's geometry
    // (job.out = window - first file byte of the window, st.zlib_size = the whole image's): bits outside the band stay 0
    const bool band = !job.whole_png;
    if (band && !(job.flags & 0x100u)) return;
    const int64_t data_begin = kPngHeaderBytes, data_end = (int64_t)(kPngHeaderBytes + st.zlib_size - 4);
    const int64_t end_aligned = (data_end + 15) & ~15ll;
    const uint32_t range_bytes = 1u << uniform(crc_range_log2(st));
    const int64_t range_end = end_aligned - (int64_t)blockIdx.x * range_bytes;
    if (range_end <= (data_begin & ~15ll)) return; // nothing of the data in this range
    const uint32_t eob_bits = job.is_last ? (job.table->lit[256] >> 16) : 0u; // only the image's last band carries the end-of-block symbol
    if (band) { // most ranges of the file belong to other bands
        const int64_t fb0 = job.is_first ? 0 : (int64_t)row_off[job.row_base] + job.bit_bias;
        const int64_t fb1 = (int64_t)st.token_end_bit + eob_bits + job.bit_bias;
        if (fb1 <= (range_end - (int64_t)range_bytes) * 8 || fb0 >= range_end * 8) return;
    }
    for (int i = threadIdx.x; i < 16 * 256; i += kBlock) (&tab[0][0])[i] = (&tabs->striped[0][0])[i];
    __syncthreads();
    // (measured and dropped: the barrier in front of the CRC steps only, so that the row search's loads travel with the staging
    // loads, and the lane's final constant loaded here -- single frames 0.3 us faster, the 8 x 8K step 1-2 % slower)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = uniform(tid >> 6);

    // Everything below is relative to the first byte of the block's range, in 32-bit arithmetic: positions that
    // matter lie within +-2^31 bits of it (a row has < 2^30 token bits), the rest saturates.
    const int64_t range_begin = range_end - range_bytes; // may be negative: bytes in front of the file count as absent
    auto sat = [](int64_t v) { return (int32_t)(v > 0x7FFFFFFFll ? 0x7FFFFFFFll : (v < -0x7FFFFFFFll ? -0x7FFFFFFFll : v)); };
    const int32_t db = sat(data_begin - range_begin), de = sat(data_end - range_begin); // bytes
    gptr_cu8 base = to_global<gptr_cu8>(job.out) + range_begin;

    const bool compressed = uniform(st.mode) == 0u;
    const bool gather = compressed; // (a band that is to be stored gathers nothing: its bytes are in place, only the CRC is taken)
    if (!compressed && job.whole_png && adler_parts) { // the image fell back to stored blocks: this workgroup writes its range of them
        const size_t slot = (size_t)blockIdx.y * max_crc_blocks + blockIdx.x;
        assemble_stored(job, st, range_begin, range_bytes, db, de, tab, red, tabs, &partials[slot], &adler_parts[2 * slot]);
        return;
    }
    const int64_t bit0 = range_begin * 8 - job.bit_bias; // zlib bit position of relative bit 0
    const uint32_t R = uniform(job.nrows);
    const uint32_t stride = uniform(job.local_stride);
    gptr_cu32 loc = to_global<gptr_cu32>(local) + job.local_base;
    const FPNG_GLOBAL uint64_t *offs = (const FPNG_GLOBAL uint64_t *)(uintptr_t)(row_off + job.row_base);
    int32_t tok_begin = 0, tok_end = 0; // first token; end of the end-of-block symbol (= end of the last row's local stream)
    // Row cursor of the WAVE (all of it wave-uniform): per step a wave covers 1 KiB of the file, row r is the
    // one holding the first token bit of that chunk and spans bits [a, n).
    uint32_t r = 0;
    int32_t a = 0, n = 0;
    uint32_t pr = 0; // look-ahead: lane l of pv = begin of row pr+1+l (tok_end past the last row), one coalesced load
    int32_t pv = 0;
    auto row_begin_lane = [&](uint32_t rr) { return (rr < R) ? sat((int64_t)offs[rr] - bit0) : tok_end; }; // per lane
    auto row_begin = [&](uint32_t rr) { return (int32_t)uniform((uint32_t)row_begin_lane(rr)); };           // uniform rr
    if (gather) {
        tok_begin = (int32_t)uniform((uint32_t)sat((int64_t)offs[0] - bit0));
        tok_end = (int32_t)uniform((uint32_t)sat((int64_t)st.token_end_bit + (int64_t)eob_bits - bit0));
        // 64-ary search for the row holding the wave's first position
        const int32_t c0 = (int32_t)(wv * 8192u);
        const int32_t p0 = c0 > tok_begin ? c0 : tok_begin;
        uint32_t lo = 0, span = R; // the answer lies in [lo, lo + span); row lo starts at or before p0
        while (span > 1) {
            const uint32_t step = (span + 63) >> 6;
            const bool probe = lane * step < span;
            const int32_t v = probe ? row_begin_lane(lo + lane * step) : 0;
            const uint32_t kcnt = (uint32_t)__popcll(__ballot(probe && v <= p0)); // >= 1: lane 0 qualifies
            lo += (kcnt - 1) * step;
            span = (span - (kcnt - 1) * step < step) ? span - (kcnt - 1) * step : step;
        }
        r = uniform(lo);
        a = row_begin(r);
        pr = r;
        pv = row_begin_lane(r + 1 + lane);
        n = __builtin_amdgcn_readlane(pv, 0);
    }
    // begin of row x > pr (wave-uniform x): out of the look-ahead vector when it is there
    auto begin_of = [&](uint32_t x) {
        const uint32_t l = x - pr - 1;
        return (l < 64) ? (int32_t)__builtin_amdgcn_readlane(pv, (int)(l & 63)) : row_begin(x);
    };

    uint32_t c = 0;
    for (uint32_t row = 0; row < range_bytes / kCrcRowBytes; row++) {
        const int32_t o = (int32_t)(row * kCrcRowBytes + tid * 16); // byte, relative
        const int32_t P = o * 8;
        const int32_t C0 = (int32_t)((row * kCrcRowBytes + wv * 1024u) * 8u), C1 = C0 + 8192; // the wave's chunk (bits)
        uint32_t w[4] = {0, 0, 0, 0};
        const bool in_data = o + 16 > db && o < de;
        if (in_data && (!gather || (P < tok_begin && job.is_first))) { // stored image, or the piece (also) holds head bytes: scan_kernel wrote them
            const u32x4 d = *(gptr_cu128)(base + o);
            w[0] = d.x, w[1] = d.y, w[2] = d.z, w[3] = d.w;
        }
        if (gather && C1 > tok_begin && C0 < tok_end && (C1 >> 3) > db && (C0 >> 3) < de) { // wave-uniform
            const int32_t cm = C0 > tok_begin ? C0 : tok_begin;
            // advance the cursor through the look-ahead vector: no memory access unless it runs out
            while (r + 1 < R && n <= cm) {
                const uint32_t first = r - pr; // lane holding row r+1
                if (first >= 64) {
                    pr = r;
                    pv = row_begin_lane(r + 1 + lane);
                    continue;
                }
                const uint32_t kcnt = (uint32_t)__popcll(__ballot(lane >= first && pr + 1 + lane < R && pv <= cm)); // >= 1
                r += kcnt;
                a = __builtin_amdgcn_readlane(pv, (int)(first + kcnt - 1));
                n = begin_of(r + 1);
            }
            gptr_cu32 src = loc + (uint64_t)r * stride;
            if (C0 >= a && C1 <= n) {
                // the whole chunk comes from one row: five dwords, four funnel shifts per lane
                const uint32_t p = (uint32_t)(P - a);
                gptr_cu32 q = src + (p >> 5);
                const uint32_t sh = p & 31u;
                const uint32_t s0 = q[0], s1 = q[1], s2 = q[2], s3 = q[3], s4 = q[4]; // (non-temporal loads / stores here: 0.206 vs 0.193 ms)
                w[0] = __builtin_amdgcn_alignbit(s1, s0, sh);
                w[1] = __builtin_amdgcn_alignbit(s2, s1, sh);
                w[2] = __builtin_amdgcn_alignbit(s3, s2, sh);
                w[3] = __builtin_amdgcn_alignbit(s4, s3, sh);
            } else {
                // rows meet inside the chunk (or it holds the stream's begin / end): every row overlapping the
                // chunk ORs its bits into the lanes it touches; dwords outside a row's stream read as zero
                // two rows per round, their loads issued together (the second one may be empty)
                int32_t aa = a, nn = n;
                for (uint32_t j = 0;; j += 2) {
                    const bool second = nn < C1 && r + 1 + j < R; // wave-uniform: another row begins inside the chunk
                    const int32_t nn2 = second ? begin_of(r + 2 + j) : nn;
                    const int32_t pA = P - aa, pB = P - nn;          // may be negative: the row starts behind this lane's piece
                    const int32_t iA = pA >> 5, iB = pB >> 5;        // floor
                    const uint32_t ndwA = ((uint32_t)(nn - aa) + 31u) >> 5, ndwB = second ? ((uint32_t)(nn2 - nn) + 31u) >> 5 : 0u;
                    gptr_cu32 srcB = src + stride;
                    // five consecutive dwords per row and lane, loaded unconditionally (an index clamped to [-4, ndw] stays
                    // inside the scratch: streams have four dwords of slack behind them and a line of padding in front of
                    // the first one) and masked afterwards: two load instructions per row instead of five predicated ones
                    uint32_t sA[5], sB[5];
                    {
                        const int32_t cA = iA < -4 ? -4 : (iA > (int32_t)ndwA ? (int32_t)ndwA : iA);
                        const int32_t cB = iB < -4 ? -4 : (iB > (int32_t)ndwB ? (int32_t)ndwB : iB);
                        gptr_cu32 qA = src + cA, qB = srcB + cB;
                        uint32_t lA[5], lB[5];
#pragma unroll
                        for (int t = 0; t < 5; t++) lA[t] = qA[t];
#pragma unroll
                        for (int t = 0; t < 5; t++) lB[t] = qB[t];
#pragma unroll
                        for (int t = 0; t < 5; t++) sA[t] = ((uint32_t)(iA + t) < ndwA) ? lA[t] : 0u;
#pragma unroll
                        for (int t = 0; t < 5; t++) sB[t] = ((uint32_t)(iB + t) < ndwB) ? lB[t] : 0u;
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; kk++)
                        w[kk] |= __builtin_amdgcn_alignbit(sA[kk + 1], sA[kk], (uint32_t)pA & 31u) |
                                 __builtin_amdgcn_alignbit(sB[kk + 1], sB[kk], (uint32_t)pB & 31u);
                    if (!second || nn2 >= C1 || r + 2 + j >= R) break;
                    aa = nn2;
                    nn = begin_of(r + 3 + j);
                    src += 2 * (uint64_t)stride;
                }
            }
            if (in_data && P + 128 > tok_begin && P < tok_end) { // (a band's window ends with the piece that holds its last bit)
                u32x4 d; // (the last data piece also covers the bytes behind the data: finalize_kernel writes those afterwards)
                d.x = w[0], d.y = w[1], d.z = w[2], d.w = w[3];
                *(gptr_u128)(uintptr_t)(base + o) = d; // (non-temporal: 0.216 vs 0.190 ms)
            }
            if (r - pr >= 32) { // refill the look-ahead early: its latency hides behind this step's CRC
                pr = r;
                pv = row_begin_lane(r + 1 + lane);
            }
        }
        if (in_data && (o < db || o + 16 > de)) { // zero the bytes outside [data_begin, data_end) for the CRC
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                uint32_t m = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int32_t pos = o + 4 * kk + b;
                    if (pos >= db && pos < de) m |= 0xFFu << (8 * b);
                }
                w[kk] &= m;
            }
        }
        if (!in_data) w[0] = w[1] = w[2] = w[3] = 0;
        w[0] ^= c;
        uint32_t nx = 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            nx ^= tab[4 * kk + 0][w[kk] & 0xFF] ^ tab[4 * kk + 1][(w[kk] >> 8) & 0xFF] ^ tab[4 * kk + 2][(w[kk] >> 16) & 0xFF] ^
                  tab[4 * kk + 3][w[kk] >> 24];
        c = nx;
    }
    c = wave_xor(dev_mulmod(c, tabs->lane_fix[tid]));
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) partials[(size_t)blockIdx.y * max_crc_blocks + blockIdx.x] = red[0] ^ red[1] ^ red[2] ^ red[3];
}

// ---------------------------------------------------------------------------------------------
// build_dynamic_kernel (2-pass): histogram -> per-image Huffman table + Deflate dynamic block
// header, one wave per image, everything in LDS.
//
// The table must be THE table the reference builds, not merely an optimal one: code lengths depend
// on its tie-breaking (reference fpng.cpp:868-907 adjust_freq32, :622-709 sort / minimum redundancy /
// length limit / canonical codes, :746-816 header).  Restated here as: stable rank sort (parallel),
// two-queue Huffman merge that prefers the leaf on ties with 16-bit node weights, leaf depths by
// parent walking (parallel), Kraft repair, shortest codes to the heaviest symbols, canonical
// bit-reversed codes, run-length packing of the code lengths with symbols 16/17/18.
// ---------------------------------------------------------------------------------------------
// HCLEN swizzle order of the code-length code's lengths (RFC 1951 3.2.7; reference fpng.cpp:728)
__device__ __constant__ uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct __attribute__((aligned(16))) BuilderLds {
    uint32_t count[288];            // 16-bit symbol counts
    uint32_t skey[288], ssym[288];  // used symbols sorted by (count, symbol)
    uint32_t iw[288];               // internal node weights (mod 2^16)
    int iparent[288], lparent[288];
    int num_codes[40];
    uint32_t len[288], code[288];   // result of the last build_table call
    uint32_t lit_len[288], lit_code[288];
    uint32_t seq[320], tok[320], ntok; // code lengths of both tables in a row; their run-length tokens (symbol | extra value << 8)
    uint32_t c2[19], cl_len[19], cl_code[19];
    uint32_t hdr[100];              // header bits, LSB-first
    uint32_t used, tmp;
};

__device__ __forceinline__ uint32_t dev_bitrev(uint32_t v, uint32_t n)
{
    return n ? (__brev(v) >> (32 - n)) : 0u;
}

// Code lengths (<= max_len) and canonical codes for n symbols with counts L.count[0..n).
__device__ __forceinline__ void dev_build_table(BuilderLds &L, uint32_t n, uint32_t max_len, uint32_t lane, uint64_t *tb = nullptr) // (inlined: L must be known to live in LDS)
{
#ifdef FPNG_BUILD_TIMING
#define FPNG_TB(i) if (tb) tb[i] = __builtin_readcyclecounter()
#else
#define FPNG_TB(i)
#endif
    FPNG_TB(0);
    // ---- stable sort of the used symbols by count: every element computes its own rank.  Composite keys
    //      count << 9 | symbol are distinct, so the rank is a plain "how many keys are smaller" (branch-free, four keys
    //      per LDS load); unused symbols get the largest key and do not take part. ----
    for (uint32_t i = lane; i < 288; i += kWave) {
        const uint32_t k = (i < n) ? L.count[i] : 0u;
        L.code[i] = k ? ((k << 9) | i) : 0xFFFFFFFFu; // (L.code doubles as key scratch: it is rebuilt at the end)
    }
    wave_lds_fence();
    {
        uint32_t own[5], rank[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 5; q++) own[q] = (lane + 64u * q < 288u) ? L.code[lane + 64u * q] : 0xFFFFFFFFu;
        const u32x4 *k4 = (const u32x4 *)L.code;
        const uint32_t n4 = (n + 3) >> 2;
#pragma unroll 2
        for (uint32_t j = 0; j < n4; j++) {
            const u32x4 v = k4[j];
#pragma unroll
            for (int q = 0; q < 5; q++) rank[q] += (uint32_t)(v.x < own[q]) + (uint32_t)(v.y < own[q]) + (uint32_t)(v.z < own[q]) + (uint32_t)(v.w < own[q]);
        }
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < 5; q++)
            if (own[q] != 0xFFFFFFFFu) {
                L.skey[rank[q]] = own[q] >> 9;
                L.ssym[rank[q]] = own[q] & 511u;
                mine++;
            }
        const uint32_t used_all = wave_sum(mine);
        if (lane == 0) L.used = used_all;
    }
    wave_lds_fence();
    for (uint32_t i = lane; i < 40; i += kWave) L.num_codes[i] = 0;
    for (uint32_t i = lane; i < n; i += kWave) L.len[i] = 0, L.code[i] = 0;
    wave_lds_fence();
    FPNG_TB(1); // rank sort
    const uint32_t used = L.used;
    if (used == 1) {
        if (lane == 0) L.num_codes[1] = 1;
    } else if (used >= 2) {
        {
            // Two-queue merge: leaves ascending, internal nodes in creation order; an internal node is taken only when strictly
            // lighter than the next leaf (reference fpng.cpp:645-651).  The reference takes its 2 * (used - 1) picks one after the
            // other; here the WAVE takes a run of picks at once wherever the choices of the run do not depend on what the run makes:
            //   * leaves: while the queue of internal nodes does not move, its head H is fixed -- every leaf of the run with
            //     !(H < key) is taken, two by two, lane l making node made + l (with the queue empty the first pair only: it
            //     becomes the head the next picks are compared with);
            //   * internal nodes that exist already: while no leaf is taken, the next leaf K is fixed -- every node of the run with
            //     weight < K (all of them once the leaves are used up) is taken, two by two;
            //   * anything else (a leaf and a node, a run of one): one step as the reference does it.
            // Histograms of images begin with runs of symbols that occur a few times (the 8K `grad` frame: 262 symbols, 13 steps
            // of the wave + 8 single ones instead of 261; a photograph: 85 + 102) -- tests/cpp/merge_model.cpp holds the two forms
            // against each other (keys whose sums wrap at 16 bits included: every comparison is the reference's own).
            uint32_t leaf = 0, root = 0, made = 0;
            while (made + 1 < used) {
                const bool q = root < made, have_leaf = leaf < used;
                const uint32_t H = uniform(q ? L.iw[root] : 0u), K = uniform(have_leaf ? L.skey[leaf] : 0u);
                uint32_t pairs = 0;
                if (have_leaf && !(q && H < K)) {
                    const uint32_t li = leaf + lane;
                    const bool c = li < used && !(q && H < L.skey[li < used ? li : leaf]);
                    const uint64_t stop = ~__ballot(c);
                    uint32_t run = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                    if (!q && run > 2) run = 2;
                    pairs = run >> 1;
                    if (lane < pairs) {
                        const uint32_t node = made + lane;
                        L.iw[node] = (L.skey[leaf + 2 * lane] + L.skey[leaf + 2 * lane + 1]) & 0xFFFFu;
                        L.lparent[leaf + 2 * lane] = (int)node;
                        L.lparent[leaf + 2 * lane + 1] = (int)node;
                    }
                    leaf += 2 * pairs;
                } else {
                    const uint32_t ri = root + lane;
                    const bool c = ri < made && (!have_leaf || L.iw[ri < made ? ri : root] < K);
                    const uint64_t stop = ~__ballot(c);
                    const uint32_t run = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                    pairs = run >> 1;
                    if (lane < pairs) {
                        const uint32_t node = made + lane;
                        L.iw[node] = (L.iw[root + 2 * lane] + L.iw[root + 2 * lane + 1]) & 0xFFFFu;
                        L.iparent[root + 2 * lane] = (int)node;
                        L.iparent[root + 2 * lane + 1] = (int)node;
                    }
                    root += 2 * pairs;
                }
                made += pairs;
                if (!pairs) { // one step of the reference's loop (all lanes walk through it; lane 0 writes)
                    uint32_t wsum = 0;
                    for (int k = 0; k < 2; k++) {
                        const uint32_t iw_head = uniform(root < made ? L.iw[root] : 0u), key_head = uniform(leaf < used ? L.skey[leaf] : 0u);
                        if (leaf >= used || (root < made && iw_head < key_head)) {
                            wsum += iw_head;
                            if (lane == 0) L.iparent[root] = (int)made;
                            root++;
                        } else {
                            wsum += key_head;
                            if (lane == 0) L.lparent[leaf] = (int)made;
                            leaf++;
                        }
                    }
                    if (lane == 0) L.iw[made] = wsum & 0xFFFFu;
                    made++;
                }
                wave_lds_fence();
            }
        }
        wave_lds_fence();
        FPNG_TB(2); // merge
        // leaf depth = number of parent hops to the root (the last internal node)
        for (uint32_t i = lane; i < used; i += kWave) {
            int node = L.lparent[i], d = 1;
            while (node != (int)used - 2) {
                node = L.iparent[node];
                d++;
            }
            atomicAdd(&L.num_codes[d > 39 ? 39 : d], 1);
        }
        wave_lds_fence();
        FPNG_TB(3); // depths
        {
            // Kraft repair (reference fpng.cpp:663-674), by the whole wave with lane l holding num_codes[l]: the
            // reference's inner search "largest l < max_len with codes" is one ballot instead of a walk through LDS
            // (skewed histograms need hundreds of repair steps).
            int nc = (lane < 40) ? L.num_codes[lane] : 0;
            const uint32_t over = wave_sum((lane > max_len && lane < 40) ? (uint32_t)nc : 0u);
            if (lane == max_len) nc += (int)over;
            if (lane > max_len) nc = 0;
            uint32_t total = wave_sum((lane >= 1 && lane <= max_len) ? ((uint32_t)nc << (max_len - lane)) : 0u);
            // The reference's loop moves one code per iteration, depth-first: a code taken from level l (the largest level below max_len
            // that has codes) is split down to max_len -- 2^(max_len - l) - 1 iterations, after which the levels between are empty again
            // and level max_len has gained one code -- before the next code of level l is touched.  As many whole walks as the sum is
            // still too large by are applied in one go; what is left over takes single iterations, at most one per level (`grad`: 13
            // passes through this loop instead of 256).  tests/cpp/merge_model.cpp holds this form against the reference's.
            while (total != (1u << max_len)) {
                const uint64_t have = __ballot(nc != 0 && lane >= 1 && lane < max_len);
                const uint32_t excess = total - (1u << max_len);
                if (have) {
                    const uint32_t l = 63u - (uint32_t)__builtin_clzll(have);
                    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane(nc, (int)l), cost = (1u << (max_len - l)) - 1u;
                    const uint32_t afford = excess / cost, k = cnt < afford ? cnt : afford;
                    if (k) {
                        if (lane == l) nc -= (int)k;
                        if (lane == max_len) nc += (int)k;
                        total -= k * cost;
                    } else {
                        if (lane == max_len) nc--;
                        if (lane == l) nc--;
                        if (lane == l + 1) nc += 2;
                        total--;
                    }
                } else {
                    if (lane == max_len) nc--;
                    total--;
                }
            }
            if (lane < 40) L.num_codes[lane] = nc;
        }
    }
    wave_lds_fence();
    FPNG_TB(4); // Kraft
    // shortest codes to the END of the sorted order (reference fpng.cpp:697-698)
    for (uint32_t i = lane; i < used; i += kWave) {
        const uint32_t from_end = used - 1 - i;
        uint32_t acc = 0, l = 1;
        for (; l <= max_len; l++) {
            acc += (uint32_t)L.num_codes[l];
            if (acc > from_end) break;
        }
        L.len[L.ssym[i]] = l;
    }
    wave_lds_fence();
    // canonical codes in symbol order, bit-reversed (reference fpng.cpp:699-708): a symbol's code = first code of its
    // length + number of earlier symbols of the same length (ballot prefix counts, one length at a time)
    {
        uint32_t len5[5];
#pragma unroll
        for (int q = 0; q < 5; q++) len5[q] = (lane + 64u * q < n) ? L.len[lane + 64u * q] : 0u;
        const uint64_t lt_mask = (1ull << lane) - 1ull;
        uint32_t first = 0;
        for (uint32_t l = 1; l <= max_len; l++) {
            uint32_t seen = 0; // symbols of length l in the rounds before
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const uint64_t m = __ballot(len5[q] == l);
                if (len5[q] == l) L.code[lane + 64u * q] = dev_bitrev(first + seen + (uint32_t)__popcll(m & lt_mask), l);
                seen += (uint32_t)__popcll(m);
            }
            first = (first + uniform((uint32_t)L.num_codes[l])) << 1;
        }
    }
    wave_lds_fence();
    FPNG_TB(5); // lengths + codes
}

__global__ __launch_bounds__(kWave) void build_dynamic_kernel(const Job *jobs, uint32_t *hist_all, TokenTable *tables, uint32_t rezero)
{
    __shared__ BuilderLds L;
    const Job &job = jobs[blockIdx.x];
    const uint32_t lane = threadIdx.x, c = job.c;
    uint32_t *hist = hist_all + (size_t)blockIdx.x * 288;
    const TokenTable *symtab = job.table; // chunk[q] = (length symbol - 256) | extra_bits << 8 | extra_value << 16
    TokenTable *out = tables + blockIdx.x;

    // ---- adjust_freq32 (reference fpng.cpp:868-907): scale to 16 bits, never to zero.  Its
    //      ">65535" repair loop only rewrites the 32-bit input, which nothing reads afterwards. ----
    // (table training, job flag 0x400: the corpus sums come with their own end-of-block count, reference fpng.cpp:932-954;
    // an image's histogram has it set to 1, :1092 / :1371)
    const bool corpus = (job.flags & 0x400u) != 0;
    uint32_t part = 0;
    for (uint32_t i = lane; i < 288; i += kWave) part += (i == 256 && !corpus) ? 1u : hist[i];
    const uint32_t total = wave_sum(part); // uint32 wrap-around like the reference's total_freq
    for (uint32_t i = lane; i < 288; i += kWave) {
        const uint32_t f = (i == 256 && !corpus) ? 1u : hist[i];
        uint32_t v = 0;
        if (f && total) {
            v = (uint32_t)(((uint64_t)f * 65535ull) / total);
            if (!v) v = 1;
        }
        if (i == 256) v = 1; // reference fpng.cpp:757
        L.count[i] = v;
        if (rezero) hist[i] = 0; // (read for the last time: the next submission's histogram pass finds its counters cleared)
    }
    for (uint32_t i = lane; i < 100; i += kWave) L.hdr[i] = 0;
    wave_lds_fence();
#ifdef FPNG_BUILD_TIMING
    uint64_t bt[8];
    bt[0] = __builtin_readcyclecounter();
#define FPNG_BT(i) bt[i] = __builtin_readcyclecounter()
#else
#define FPNG_BT(i)
#endif
#ifdef FPNG_BUILD_TIMING
    uint64_t tb[6] = {0, 0, 0, 0, 0, 0};
    dev_build_table(L, 288, 12, lane, tb);
#else
    dev_build_table(L, 288, 12, lane);
#endif
    FPNG_BT(1);
    for (uint32_t i = lane; i < 288; i += kWave) L.lit_len[i] = L.len[i], L.lit_code[i] = L.code[i];
    wave_lds_fence();

    // distance tree: exactly two used symbols (c-1 and c), both get 1-bit codes, the real one code 0
    // (reference fpng.cpp:1095-1099 / :1375-1377)
    uint32_t n_lit = 286, n_dist = c + 1;
    while (n_lit > 257 && !L.lit_len[n_lit - 1]) n_lit--;
    const uint32_t n_seq = n_lit + n_dist;
    for (uint32_t i = lane; i < n_seq; i += kWave)
        L.seq[i] = (i < n_lit) ? L.lit_len[i] : ((i - n_lit == c - 1 || i - n_lit == c) ? 1u : 0u);
    for (uint32_t i = lane; i < 19; i += kWave) L.c2[i] = 0;
    wave_lds_fence();

    FPNG_BT(2);
    // ---- run-length packing of the code lengths (reference fpng.cpp:711-726, :770-794).  The reference walks the sequence with a
    //      small state machine (pending zero run, pending repeat count); per MAXIMAL RUN of equal lengths its output has a closed
    //      form (checked against the state machine on 200 000 random sequences, and by the 2-pass parity tests):
    //        L zeros:      L / 138 times (18, 127), then for the rest r: nothing (0) / r zeros (< 3) / (17, r - 3) (<= 10) / (18, r - 11)
    //        L lengths v:  the literal v, then (L - 1) / 6 times (16, 3), then for the rest r: r literals v (< 3) / (16, r - 3)
    //      One lane per run (five rounds cover the sequence), a scan of the runs' token counts, every lane writes its run's tokens
    //      (symbol | extra value << 8) and adds its symbols to the code-length histogram.  (The serial walk by one lane took 93 000
    //      cycles of the builder's 335 000: every step waited for an LDS round trip.) ----
    {
        uint32_t rv[5], rl[5], cnt[5], off[5];
        uint64_t sm[5];
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const uint32_t i = lane + 64u * q;
            rv[q] = (i < n_seq) ? L.seq[i] : 0u;
            sm[q] = __ballot(i < n_seq && (i == 0 || L.seq[i ? i - 1 : 0] != rv[q])); // runs start here
        }
        uint32_t base = 0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            const uint32_t i = lane + 64u * q;
            // the next run's start: the next set bit above this lane, else the first one of a later round, else the sequence's end
            uint32_t nxt = n_seq;
            bool found = false;
            const uint64_t above = sm[q] & ~((2ull << lane) - 1ull);
            if (above) nxt = 64u * q + (uint32_t)__builtin_ctzll(above), found = true;
#pragma unroll
            for (int q2 = q + 1; q2 < 5; q2++)
                if (!found && sm[q2]) nxt = 64u * q2 + (uint32_t)__builtin_ctzll(sm[q2]), found = true;
            rl[q] = ((sm[q] >> lane) & 1ull) ? nxt - i : 0u;
            uint32_t n = 0;
            if (rl[q]) {
                if (rv[q]) {
                    const uint32_t R = rl[q] - 1u, r = R % 6u;
                    n = 1u + R / 6u + (r < 3u ? r : 1u);
                } else {
                    const uint32_t r = rl[q] % 138u;
                    n = rl[q] / 138u + (r == 0u ? 0u : (r < 3u ? r : 1u));
                }
            }
            cnt[q] = n;
            const uint32_t incl = wave_inclusive_sum(n);
            off[q] = base + incl - n;
            base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
#pragma unroll
        for (int q = 0; q < 5; q++) {
            if (!rl[q]) continue;
            uint32_t o = off[q];
            const uint32_t v = rv[q];
            if (v) {
                const uint32_t R = rl[q] - 1u, full = R / 6u, r = R % 6u;
                L.tok[o++] = v;
                for (uint32_t k = 0; k < full; k++) L.tok[o++] = 16u | (3u << 8);
                if (r < 3u) {
                    for (uint32_t k = 0; k < r; k++) L.tok[o++] = v;
                } else
                    L.tok[o++] = 16u | ((r - 3u) << 8);
                atomicAdd(&L.c2[v], 1u + (r < 3u ? r : 0u));
                if (full + (r >= 3u ? 1u : 0u)) atomicAdd(&L.c2[16], full + (r >= 3u ? 1u : 0u));
            } else {
                const uint32_t full = rl[q] / 138u, r = rl[q] % 138u;
                for (uint32_t k = 0; k < full; k++) L.tok[o++] = 18u | (127u << 8);
                if (r == 0u) {
                } else if (r < 3u) {
                    for (uint32_t k = 0; k < r; k++) L.tok[o++] = 0u;
                    atomicAdd(&L.c2[0], r);
                } else if (r <= 10u) {
                    L.tok[o++] = 17u | ((r - 3u) << 8);
                    atomicAdd(&L.c2[17], 1u);
                } else {
                    L.tok[o++] = 18u | ((r - 11u) << 8);
                }
                if (full + (r > 10u ? 1u : 0u)) atomicAdd(&L.c2[18], full + (r > 10u ? 1u : 0u));
            }
        }
        if (lane == 0) L.ntok = base;
    }
    wave_lds_fence();
    for (uint32_t i = lane; i < 288; i += kWave) L.count[i] = (i < 19) ? (L.c2[i] & 0xFFFFu) : 0u;
    wave_lds_fence();
    FPNG_BT(3);
    dev_build_table(L, 19, 7, lane);
    FPNG_BT(4);
    for (uint32_t i = lane; i < 19; i += kWave) L.cl_len[i] = L.len[i], L.cl_code[i] = L.code[i];
    wave_lds_fence();

    // ---- header bits (reference fpng.cpp:1279-1283 zlib header + BFINAL, :796-813 block header): every field ORs itself into
    //      place (LDS atomics on the zeroed buffer): lane 0 the 33 fixed bits, lanes 0..nbl-1 the code-length code's lengths in
    //      swizzle order, then one lane per token behind a scan of the tokens' bit counts ----
    {
        auto put = [&](uint32_t pos, uint32_t v, uint32_t nbits) {
            const uint32_t d = pos >> 5, sh = pos & 31u;
            atomicOr(&L.hdr[d], v << sh);
            if (sh + nbits > 32u) atomicOr(&L.hdr[d + 1], v >> (32u - sh));
        };
        const uint32_t ord = kClOrder[lane < 19 ? lane : 0];
        const uint64_t nz = __ballot(lane < 19 && L.cl_len[ord] != 0);
        uint32_t nbl = nz ? 64u - (uint32_t)__builtin_clzll(nz) : 0u;
        nbl = nbl < 4u ? 4u : nbl;
        if (lane == 0) {
            const uint64_t fixed = 0x78ull | (0x01ull << 8) | (1ull << 16) | (2ull << 17) | ((uint64_t)(n_lit - 257u) << 19) |
                                   ((uint64_t)(n_dist - 1u) << 24) | ((uint64_t)(nbl - 4u) << 29);
            atomicOr(&L.hdr[0], (uint32_t)fixed);
            atomicOr(&L.hdr[1], (uint32_t)(fixed >> 32));
        }
        if (lane < nbl) put(33u + 3u * lane, L.cl_len[ord], 3u);
        uint32_t pos0 = 33u + 3u * nbl;
        const uint32_t nt = L.ntok;
        for (uint32_t t0 = 0; t0 < nt; t0 += kWave) {
            const uint32_t t = t0 + lane;
            uint32_t v = 0, nb = 0;
            if (t < nt) {
                const uint32_t tk = L.tok[t], sy = tk & 0xFFu;
                nb = L.cl_len[sy];
                v = L.cl_code[sy] | ((tk >> 8) << nb);
                nb += (sy == 16u) ? 2u : (sy == 17u ? 3u : (sy == 18u ? 7u : 0u));
            }
            const uint32_t incl = wave_inclusive_sum(nb);
            if (nb) put(pos0 + incl - nb, v, nb);
            pos0 += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) L.tmp = pos0;
    }
    wave_lds_fence();

    FPNG_BT(5);
    // ---- publish the table in the layout the row kernels consume ----
    const uint32_t hbits = L.tmp;
    for (uint32_t i = lane; i < 288; i += kWave) out->lit[i] = L.lit_code[i] | (L.lit_len[i] << 16);
    const uint32_t cap = (c == 4) ? kMaxChunkPixels4 : kMaxChunkPixels3;
    for (uint32_t q = lane; q < 96; q += kWave) {
        uint32_t v = 0;
        if (q >= 1 && q <= cap) {
            const uint32_t e = symtab->chunk[q], sy = 256 + (e & 0xFF), extra = (e >> 8) & 0xFF, ev = e >> 16;
            v = (L.lit_code[sy] | (ev << L.lit_len[sy])) | ((L.lit_len[sy] + extra + 1u) << 24);
        }
        out->chunk[q] = v;
    }
    for (uint32_t i = lane; i < 100; i += kWave) ((uint32_t *)out->header)[i] = L.hdr[i];
    if (lane == 0) {
        out->first_token_bit = hbits;
        out->header_bits = hbits;
    }
#ifdef FPNG_BUILD_TIMING
    FPNG_BT(6);
    if (lane == 0 && blockIdx.x == 0) // cycles: table(288), sequence, run-length packing, table(19), header, publish
    {
        for (int i = 0; i < 6; i++) ((uint32_t *)hist_all)[i] = (uint32_t)(bt[i + 1] - bt[i]);
        for (int i = 0; i < 5; i++) ((uint32_t *)hist_all)[8 + i] = (uint32_t)(tb[i + 1] - tb[i]); // inside table(288): sort, merge, depths, Kraft, codes
    }
#endif
}

// Table training, per image of the corpus (one wave each): its histogram adjusted to 16 bits as it enters the reference's block
// writer (fpng.cpp:868-907 with lit_freq[256] = 1; summed at :751-755, BEFORE the end-of-block count is forced to 1).
__global__ __launch_bounds__(kWave) void train_accumulate_kernel(const uint32_t *hist_all, unsigned long long *sums)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t *hist = hist_all + (size_t)blockIdx.x * 288;
    uint32_t part = 0;
    for (uint32_t i = lane; i < 288; i += kWave) part += (i == 256) ? 1u : hist[i];
    const uint32_t total = wave_sum(part);
    for (uint32_t i = lane; i < 288; i += kWave) {
        const uint32_t f = (i == 256) ? 1u : hist[i];
        if (!f || !total) continue;
        const uint32_t v = (uint32_t)(((uint64_t)f * 65535ull) / total);
        atomicAdd(&sums[i], (unsigned long long)(v ? v : 1u));
    }
}

__global__ void or_piece_kernel(uint32_t *dst, const uint32_t *src) { dst[threadIdx.x] |= src[threadIdx.x]; }

} // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static dim3 row_grid(uint32_t max_rows, uint32_t n_jobs)
{
    return dim3((max_rows + kRowWaves - 1) / kRowWaves, n_jobs, 1);
}

void launch_hist(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_rows, uint32_t *hist)
{
    hipLaunchKernelGGL(hist_kernel, dim3((max_rows + kHistWaves - 1) / kHistWaves, n_jobs, 1), dim3(kHistBlock), 0, s, jobs, hist);
}
void launch_hist_first(hipStream_t s, const Job &job, Job *d_job, uint32_t *hist)
{
    JobArg arg;
    arg.job = job;
    hipLaunchKernelGGL(hist_first_kernel, dim3((job.nrows + kHistWaves - 1) / kHistWaves, 1, 1), dim3(kHistBlock), 0, s, arg, d_job, hist);
}
void launch_scan(hipStream_t s, const Job *jobs, uint32_t n_jobs, const RowInfo *rows, uint64_t *row_off, JobState *states)
{
    hipLaunchKernelGGL(scan_kernel, dim3(n_jobs), dim3(kScanBlock), 0, s, jobs, rows, row_off, states);
}
void launch_build_dynamic(hipStream_t s, const Job *jobs, uint32_t n_jobs, const uint32_t *hist, TokenTable *tables, uint32_t rezero)
{
    hipLaunchKernelGGL(build_dynamic_kernel, dim3(n_jobs), dim3(kWave), 0, s, jobs, (uint32_t *)hist, tables, rezero);
}
void launch_encode_rows(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_rows, uint32_t chan_mask, RowInfo *rows,
                        JobState *states, uint32_t *local, bool wide4)
{
    if (chan_mask & 1u)
        hipLaunchKernelGGL((encode_rows_kernel<3, FPNG_ROWS_WPE>), row_grid(max_rows, n_jobs), dim3(kRowBlock), 0, s, jobs, rows, states, local);
    if ((chan_mask & 2u) && wide4)
        hipLaunchKernelGGL((encode_rows_kernel<4, FPNG_ROWS_WPE4>), row_grid(max_rows, n_jobs), dim3(kRowBlock), 0, s, jobs, rows, states, local);
    else if (chan_mask & 2u)
        hipLaunchKernelGGL((encode_rows_kernel<4, FPNG_ROWS_WPE>), row_grid(max_rows, n_jobs), dim3(kRowBlock), 0, s, jobs, rows, states, local);
}
void launch_encode_rows_first(hipStream_t s, const Job &job, Job *d_job, RowInfo *rows, JobState *states, uint32_t *local)
{
    JobArg arg;
    arg.job = job;
    if (job.c == 3)
        hipLaunchKernelGGL((encode_rows_first_kernel<3, FPNG_ROWS_WPE>), row_grid(job.nrows, 1), dim3(kRowBlock), 0, s, arg, d_job, rows, states, local);
    else if (job.w >= kWideRowPixels)
        hipLaunchKernelGGL((encode_rows_first_kernel<4, FPNG_ROWS_WPE4>), row_grid(job.nrows, 1), dim3(kRowBlock), 0, s, arg, d_job, rows, states, local);
    else
        hipLaunchKernelGGL((encode_rows_first_kernel<4, FPNG_ROWS_WPE>), row_grid(job.nrows, 1), dim3(kRowBlock), 0, s, arg, d_job, rows, states, local);
}
void launch_assemble(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_crc_blocks, JobState *states,
                     const uint64_t *row_off, const uint32_t *local, const CrcDeviceTables *tabs, uint32_t *partials, uint32_t *adler_parts)
{
    hipLaunchKernelGGL(assemble_kernel, dim3(max_crc_blocks, n_jobs), dim3(kBlock), 0, s, jobs, states, row_off, local, tabs,
                       partials, adler_parts, max_crc_blocks);
}
void launch_crc(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_crc_blocks, const JobState *states,
                const CrcDeviceTables *tabs, uint32_t *partials)
{
    hipLaunchKernelGGL(crc_kernel, dim3(max_crc_blocks, n_jobs), dim3(kBlock), 0, s, jobs, states, tabs, partials,
                       max_crc_blocks);
}
void launch_finalize(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_crc_blocks, const RowInfo *rows,
                     JobState *states, const CrcDeviceTables *tabs, const uint32_t *partials, const uint32_t *adler_parts, Result *results)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(n_jobs), dim3(kBlock), 0, s, jobs, rows, states, tabs, partials, adler_parts, max_crc_blocks,
                       results);
}

void launch_train_accumulate(hipStream_t s, const uint32_t *hist_all, uint32_t n_images, uint64_t *sums)
{
    hipLaunchKernelGGL(train_accumulate_kernel, dim3(n_images), dim3(kWave), 0, s, hist_all, (unsigned long long *)sums);
}
void launch_or_piece(hipStream_t s, uint8_t *dst, const uint8_t *src)
{
    hipLaunchKernelGGL(or_piece_kernel, dim3(1), dim3(4), 0, s, (uint32_t *)dst, (const uint32_t *)src);
}

} // namespace fpng_amd
