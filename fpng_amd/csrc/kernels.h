// kernels.h -- host/device shared structures and kernel launchers (internal).
#pragma once
#include "format.h"

#include <hip/hip_runtime_api.h>

namespace fpng_amd {

// CRC work decomposition: every block owns kCrcRangeBytes of the (16-byte aligned) file span,
// 256 lanes x 16 B = one 4 KiB block row per step.
constexpr uint32_t kCrcRangeBytes = 1u << 16;
constexpr uint32_t kCrcRowBytes = 4096;

// One unit of work: a whole image, or a band of rows of one image (multi-GPU).
struct Job {
    const uint8_t *rows;      // first row of the job (row y0 of the image)
    const uint8_t *row_above; // row y0-1, used only when y0 > 0
    uint8_t *out;             // whole_png: start of the .png file; band: start of the band's byte window
    const TokenTable *table;  // device table (1-pass static or per-job dynamic)
    uint64_t out_cap;
    uint64_t start_bit;       // band: absolute zlib bit of the band's first token (ignored if is_first)
    int64_t bit_bias;         // destination bit = zlib bit + bit_bias  (whole_png: 58*8)
    uint32_t w, c, bpl, nrows, y0, h_total, flags;
    uint32_t row_base;        // index of the job's first row in the per-row scratch arrays
    uint32_t one_pass, whole_png, is_first, is_last;
    uint32_t crc_blocks;      // upper bound of CRC ranges for this job
    uint64_t band_zlib_size;  // row bands, placement phase: size of the WHOLE image's zlib stream (bands share the file's geometry)
    // whole images: every row is first encoded into its own dword-aligned stream in scratch ("local stream"),
    // assemble_kernel then shifts the streams into place.  Row r lives at local + local_base + r*local_stride.
    uint64_t local_base;      // dwords
    uint32_t local_stride;    // dwords per row (worst-case token bits of a row + slack)
    uint32_t local_pad;
    uint32_t reserved[6];     // (keeps the record at 224 bytes and png_header where it was: the row walk's and the histogram pass's code is, instruction for
                              //  instruction, the build round 5's profiles were measured on -- tools/isa_diff.py)
    uint8_t png_header[60];   // 58 bytes used (reference fpng.cpp:1767-1791), IDAT length patched on device
};

struct RowInfo {
    uint32_t bits; // token bits of the row
    uint32_t s1;   // Adler raw sums of the filtered row mod 65521
    uint32_t s2;
    uint32_t pad;
};

struct JobState {
    uint64_t token_end_bit; // zlib bit position after the last token
    uint64_t zlib_size;     // bytes of the zlib stream incl. Adler
    uint32_t mode;          // 0 compressed, 1 stored
    uint32_t last_unit_bits;
    uint32_t s1, s2;        // Adler raw sums of the job's filtered bytes
    uint32_t adler, crc;
    uint32_t status;        // nonzero = device-side failure, reported in fpng_amd_result.status
    uint32_t range_log2;    // log2 of the bytes one assemble/crc block covers (12..16; 0 = 16), chosen by scan_kernel
    uint64_t reserved[2];
};

struct Result {
    uint64_t png_size;
    uint32_t mode;
    uint32_t status;
};

// device-side CRC constants (superset of CrcTables in format.h)
struct CrcDeviceTables {
    uint32_t striped[16][256]; // raw CRC of byte b, followed by (15-k) + (kCrcRowBytes-16) zero bytes
    uint32_t lane_fix[256];    // x^(8*(kCrcRowBytes - 16*tid)): lane stripe -> one row past the range end
    uint32_t pow2[48];         // x^(8*2^i)
    uint32_t inv_pad[16];      // x^(-8*p)
    uint32_t inv_row;          // x^(-8*kCrcRowBytes)
    uint32_t pad[15];
    uint32_t inv_row_pad[16];  // x^(-8*(kCrcRowBytes + p)): inv_row * inv_pad[p]
    uint32_t fold[13][256];    // fold[e-12][t] = x^(8 * 2^e * t), e = 12..24: thread t's group of partials -> the common end point
    uint32_t pow_byte[6][256]; // pow_byte[k][b] = x^(8 * b * 256^k): x^(8n) is the product of six entries
};
void build_crc_device_tables(CrcDeviceTables *t);

void launch_hist(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_rows, uint32_t *hist);
void launch_scan(hipStream_t s, const Job *jobs, uint32_t n_jobs, const RowInfo *rows, uint64_t *row_off, JobState *states);
// chan_mask: bit 0 = the batch has 3-channel jobs, bit 1 = 4-channel jobs (one kernel instantiation each)
// wide4: the 4-channel jobs' pixels lie mostly in rows of kWideRowPixels and more (their walk then runs with seven waves per SIMD instead of eight: kernels.hip)
constexpr uint32_t kWideRowPixels = 3584; // (same-box A/B: 3840-pixel rows gain with six waves, 3072-pixel rows lose 1 % in 1-pass)
void launch_encode_rows(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_rows, uint32_t chan_mask, RowInfo *rows,
                        JobState *states, uint32_t *local, bool wide4 = false);
// one job per submission: the record travels in the kernel arguments and is left at d_job for the kernels that follow
void launch_encode_rows_first(hipStream_t s, const Job &job, Job *d_job, RowInfo *rows, JobState *states, uint32_t *local);
void launch_hist_first(hipStream_t s, const Job &job, Job *d_job, uint32_t *hist);
// adler_parts (whole images; two words per CRC range and job, or NULL for row bands): where the workgroups of an image that
// fell back to stored blocks leave their range's share of the Adler-32
void launch_assemble(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_crc_blocks, JobState *states,
                     const uint64_t *row_off, const uint32_t *local, const CrcDeviceTables *tabs, uint32_t *partials, uint32_t *adler_parts);
void launch_crc(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_crc_blocks, const JobState *states,
                const CrcDeviceTables *tabs, uint32_t *partials);
void launch_finalize(hipStream_t s, const Job *jobs, uint32_t n_jobs, uint32_t max_crc_blocks, const RowInfo *rows,
                     JobState *states, const CrcDeviceTables *tabs, const uint32_t *partials, const uint32_t *adler_parts, Result *results);
// table training: sums[0..288) += the 16-bit adjusted histogram of every image (hist_all: 288 counters per image)
void launch_train_accumulate(hipStream_t s, const uint32_t *hist_all, uint32_t n_images, uint64_t *sums);
// dst[0..16) |= src[0..16): the 16-byte piece two neighbouring band windows share (each holds zeros where the other's bits are)
void launch_or_piece(hipStream_t s, uint8_t *dst, const uint8_t *src);
// rezero: leave the counters that were read zeroed (the next 2-pass submission's histogram pass needs no clearing in front of it)
void launch_build_dynamic(hipStream_t s, const Job *jobs, uint32_t n_jobs, const uint32_t *hist, TokenTable *tables, uint32_t rezero);

} // namespace fpng_amd
