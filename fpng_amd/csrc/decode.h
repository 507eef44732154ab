// decode.h -- host/device shared structures of the GPU batch decoder (decode.hip, decode_api.cpp).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace fpng_amd {

constexpr uint32_t kSubBits = 512;      // token bits per subsequence (one thread each)
constexpr uint32_t kDecSubBlock = 512;  // subsequences per workgroup of the synchronisation (a file's subsequences are padded to whole workgroups)
#ifndef FPNG_DEC_LEADIN // (build variants lead96 / lead64: a shorter lead-in is less work for every thread and more threads to correct)
#define FPNG_DEC_LEADIN 128
#endif
constexpr uint32_t kDecLeadIn = FPNG_DEC_LEADIN; // bits a subsequence's first decode starts early (decode_core.h: sub_first); a multiple of 32
#ifndef FPNG_DEC_UNF_ROWS
#define FPNG_DEC_UNF_ROWS 48
#endif
constexpr uint32_t kDecUnfRows = FPNG_DEC_UNF_ROWS; // rows per segment of the Up filter's undoing (held in registers)
enum : uint32_t { kDecNotConverged = 1u, kDecBadStream = 2u, kDecBadFilter = 8u, kDecStalled = 16u, kDecStoredOdd = 32u, kDecSawEob = 0x100u };

struct DecJob {
    const uint8_t *z;         // device: the zlib stream (IDAT payload) from the dword its first byte (0x78) lies in; readable up to z_bytes + 16 rounded down to a dword
    uint64_t z_bytes;         // length of the IDAT payload + z_shift
    uint64_t first_bit;       // first row token (behind the dynamic block header)
    uint64_t end_limit_bit;   // (z_bytes - 4) * 8: no token may start here or later
    const uint32_t *lut;      // device: dec::kLutDwords words (decode_core.h)
    uint32_t *win;            // device scratch: h x dec_col_blocks() x dec::kWinWords words: for each window of dec_unfilter_kernel's tiles (decode_core.h:
                              // Window) the file's subsequence (counted from its first one) in whose output it begins, 0xFFFFFFFF: none; and where
                              // that subsequence's walk may begin (decode_core.h: Resume)
    uint8_t *out;             // device: w * h * dst_c pixels
    uint32_t *segsum;         // device scratch: nseg x ceil(bpl / 4) 8-byte granules {tag, column sum} of dec_unfilter_kernel's look-back
    uint32_t w, h, src_c, dst_c, bpl;
    uint32_t n_sub;           // subsequences of the file
    uint32_t sub_base;        // index of its first subsequence (a multiple of kDecSubBlock: one file per workgroup)
    uint32_t mode;            // 0 one dynamic block, 1 stored blocks
    uint32_t nseg;            // segments of kDecUnfRows rows (dec_unfilter_*_kernel)
    uint32_t pad_[2];
    uint32_t z_shift;         // bytes between z (rounded down to a dword) and the stream's first byte; the bit positions above count from z
};

// Column blocks of dec_unfilter_kernel for one file: 256 PIXELS of the file's rows each -- 256 dword columns of 4-byte pixels, or
// 4 waves x 48 dword columns of 3-byte ones (see the kernel) -- so that a match, which repeats whole pixels, is whole pixels in
// every block it touches.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t dec_col_blocks(uint32_t w, uint32_t /*src_c*/, uint32_t /*dst_c*/) { return (w + 255u) / 256u; }

// bytes of a row that one column block covers (the first one also holds the row's filter byte: decode_core.h, Window)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t dec_col_block_bytes(uint32_t src_c, uint32_t /*dst_c*/) { return 256u * src_c; }

// what the synchronisation leaves per workgroup of kDecSubBlock subsequences (indices inside the workgroup, kDecSubBlock = none)
struct DecBlockRec {
    uint32_t sum;           // output bytes of its subsequences
    uint32_t first_eob;     // first one that met an end-of-block symbol
    uint32_t first_invalid; // first one whose decode derailed
    uint32_t first_overflow; // first one that needed more records than it has room for (decode_core.h: kRecCap)
    uint32_t entry_rel;     // where its first subsequence starts, in bits behind the workgroup's first nominal bit
    uint32_t exit_rel;      // where its last subsequence ends, in bits behind the next workgroup's first nominal bit
    uint32_t want_rel;      // dec_chain_kernel: where its first subsequence must start (kDecWantUnknown: ask the workgroup in front)
    uint32_t map[3];        // dec::PhaseMap: what the workgroup does to the phases it can be entered in (entry -> exit: one pair unless the
                            // stream is periodic, decode_core.h; no pair at all: round 0 left the workgroup unsettled)
    uint32_t left;          // ... and why (decode.hip: many threads to correct / corrections that do not end)
};
constexpr uint32_t kDecWantUnknown = 0xFFFFFFFFu;

// dec_unfilter_kernel's work items -- (segment of rows, block of 256 dword columns) of a file -- numbered segment by segment over a
// group of files: the files sorted by segment count (most first) in order[]; cbpre[k] = column blocks of the first k of them;
// a piece = a range of segments over which the same `alive` first files of that order still have rows
struct DecUnfPiece {
    uint32_t item0; // number of its first item
    uint32_t seg0;  // its first segment
    uint32_t alive;
    uint32_t pad_;
};
struct DecUnfPlan {
    const DecUnfPiece *pieces;
    const uint32_t *cbpre; // n_files + 1
    const uint32_t *order; // n_files: indices into the group's jobs
    uint32_t n_pieces, total_items;
    uint32_t n_files, pad_; // files of the order (cbpre holds n_files + 1 words)
};

// a device-resident file (dec_fetch_kernel gathers the heads and tails of a batch's files for the host's container walk)
struct DecFileRef {
    const uint8_t *data;
    uint32_t size, pad_;
};
// sizes: n x 288 code lengths (checked by the host) -> luts: n x FPNG_AMD_DECODE_LUT_WORDS
void launch_dec_build_luts(hipStream_t s, const uint8_t *sizes, uint32_t n, uint32_t *luts);
void launch_dec_fetch(hipStream_t s, const DecFileRef *files, uint32_t n, uint32_t head, uint32_t tail, uint8_t *out);

// a file that is decoded piece by piece: what the pieces so far amount to (dec_offsets_range_kernel)
struct DecCarry {
    uint64_t bytes; // output bytes of the blocks so far (up to the end-of-block symbol once it was met)
    uint32_t done;  // the stream's end-of-block symbol was met
    uint32_t pad_;
};

// per-subsequence arrays (index = batch-wide subsequence number)
struct DecSubArrays {
    uint32_t *info;   // dec::pack_info
    uint32_t *bytes;  // output bytes
    uint32_t *rel;    // output offset inside its workgroup (exclusive scan of bytes)
    uint32_t *lastpx; // the four literal bytes in front of it (decode_core.h: lookback_lastpx, over the records)
    uint32_t *eob;    // where its end-of-block symbol ends, in bits behind its nominal first bit (subsequences flagged kSubEob)
    uint64_t *tok;    // its token records, two to an entry (decode_core.h: rec_index)
};

// the kernels work on the workgroups [first_block, first_block + n_blocks) of the batch's subsequences (one group of files);
// jobs / n_jobs: the whole batch
// resident: how many persistent workgroups to launch at most; changed: set by a border round that changed something; multi: set
// once a workgroup's map has more than one pair (zero at the start of the call, never cleared inside it)
void launch_dec_sync(hipStream_t s, uint32_t resident, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, uint32_t round, DecSubArrays a,
                     DecBlockRec *recs, uint32_t *changed, uint32_t *multi);
// group_jobs: the group's first file; status / eob_index: batch-wide arrays
void launch_dec_offsets(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, const DecJob *group_jobs,
                        uint32_t n_group_jobs, DecSubArrays a, const DecBlockRec *recs, uint64_t *block_off, uint32_t *status, uint32_t *eob_index);
// ONE file (jobs[0], a DEVICE pointer whose sub_base the caller knows: passed as first_block of the other launchers), blocks [blk_a, blk_b)
void launch_dec_offsets_range(hipStream_t s, const DecJob *jobs, uint32_t sub_base_block, uint32_t blk_a, uint32_t blk_b, bool final_piece, uint32_t total_subs, DecSubArrays a,
                              const DecBlockRec *recs, uint64_t *block_off, uint32_t *status, uint32_t *eob_index, DecCarry *carry);
// What the pass that writes the pixels needs of the synchronisation's results: the per-subsequence arrays, the workgroups' output offsets,
// every file's last subsequence (the one with the stream's end-of-block symbol; sub_limit: subsequences of a file that arrives in
// pieces which have been placed so far, else 0xFFFFFFFF)
struct DecPlaced {
    DecSubArrays a;
    const uint64_t *block_off;
    const uint32_t *eob_index; // per file of the launch's `jobs`
    uint32_t sub_limit;
};
// concurrent_status: kernels that may set the file's status bits run next to this launch (no workgroup may then skip its file: dec_unfilter_kernel)
void launch_dec_unfilter(hipStream_t s, const DecJob *jobs, DecUnfPlan plan, DecPlaced placed, uint32_t item0, uint32_t n_items, uint32_t *status, uint32_t epoch, bool concurrent_status);
void launch_dec_finish(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, DecUnfPlan plan, DecPlaced placed, uint32_t *status, uint32_t epoch, bool any_stored);
#ifdef FPNG_DEC_SYNC_TIMING
void dec_dump_sync_times(const char *path, uint32_t n_blocks); // (diagnostic build: dec_sync_kernel<false>'s per-workgroup time stamps of the last launch)
#endif
#ifdef FPNG_DEC_TILE_TIMING
void dec_dump_tile_times(const char *path, uint32_t n_items); // (diagnostic build: dec_unfilter_kernel's per-tile time stamps of the last launch)
#endif

} // namespace fpng_amd
