// decode.h -- host/device shared structures of the GPU batch decoder (decode.hip, decode_api.cpp).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace fpng_amd {

constexpr uint32_t kSubBits = 512; // token bits per subsequence (one thread each)
constexpr uint32_t kDecSubBlock = 512; // subsequences per workgroup of the decoding kernels (a file's subsequences are padded to whole workgroups)
constexpr uint32_t kDecUnfRows = 128; // rows per segment of the Up filter's column sums
enum : uint32_t { kDecNotConverged = 1u, kDecBadStream = 2u, kDecSawEob = 0x100u };

struct DecJob {
    const uint8_t *z;         // device: the zlib stream (16-byte aligned copy), z[0] = 0x78
    const uint8_t *z_aligned; // == z
    uint64_t z_offset;        // 0 (bit positions count from z)
    uint64_t z_bytes;         // length of the IDAT payload
    uint64_t first_bit;       // first row token (behind the dynamic block header)
    uint64_t end_limit_bit;   // (z_bytes - 4) * 8: no token may start here or later
    const uint32_t *lut;      // device: 4096 x (symbol | code length << 9 | length symbols: extra bits << 13 | base length << 16), 0 = no such code
    uint8_t *filt;            // device scratch: the filtered image, h rows of fstride bytes; a row's pixel bytes start at byte 4
                              // (dword aligned: the column kernel works on dwords), its filter byte would sit at byte 3
    uint32_t fstride;         // (bpl + 3 & ~3) + 4
    uint32_t *runmask;        // device scratch, zeroed: one bit per pixel, rows padded to 32 pixels
    uint8_t *out;             // device: w * h * dst_c pixels
    uint32_t *segsum;         // device scratch: (nseg - 1) x (fstride / 4 - 1) dwords, the Up filter's column sums per segment of rows
    uint32_t w, h, src_c, dst_c, bpl;
    uint32_t n_sub;           // subsequences of the file
    uint32_t sub_base;        // index of its first subsequence (a multiple of the block size: one file per workgroup)
    uint32_t mode;            // 0 one dynamic block, 1 stored blocks
    uint32_t nseg;            // segments of kDecUnfRows rows (dec_unfilter_*_kernel)
};

// what dec_blocksum_kernel leaves per workgroup of kDecSubBlock subsequences (indices inside the workgroup, kDecSubBlock = none)
struct DecBlockRec {
    uint32_t sum;             // output bytes of its subsequences
    uint32_t first_eob;       // first one that met an end-of-block symbol
    uint32_t first_unchained; // first one that does not start where its predecessor ended
    uint32_t first_invalid;   // first one whose decode derailed
};

// the decoding kernels work on the workgroups [first_block, first_block + n_blocks) of the batch's subsequences (one group of files);
// jobs / n_jobs: the whole batch
void launch_dec_sync(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, uint32_t round, uint64_t *start,
                     uint64_t *end, uint32_t *bytes, uint32_t *flags, uint32_t *changed);
// group_jobs / status / eob_index: of the group's first file
void launch_dec_offsets(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, const DecJob *group_jobs,
                        uint32_t n_group_jobs, const uint64_t *start, const uint64_t *end, const uint32_t *bytes, const uint32_t *flags, DecBlockRec *recs,
                        uint64_t *block_off, uint32_t *status, uint32_t *eob_index);
// status / eob_index: of the batch's first file
void launch_dec_emit(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t first_block, uint32_t n_blocks, uint32_t total_subs, const uint64_t *start,
                     const uint32_t *bytes, const uint32_t *eob_index, const uint64_t *block_off, uint32_t *status);
void launch_dec_finish(hipStream_t s, const DecJob *jobs, uint32_t n_jobs, uint32_t max_rows, uint32_t max_bpl, const uint32_t *status);

} // namespace fpng_amd
