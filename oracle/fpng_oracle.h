/*
 * fpng_oracle.h -- CPU restatement of the fpng encode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity checker for the MI355X HIP encoder.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link/load it; the product path never routes through it.
 *
 * Parity status: PINNED.  The restatement is checked byte-for-byte against the unmodified
 * reference (oracle/_ref/libfpng_ref.so, built from /root/reference/src/fpng.cpp by
 * oracle/Makefile) by tests/test_oracle_vs_ref.py, and against the committed golden vectors in
 * tests/golden/ (generated from the reference by oracle/make_golden.py).
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 */
#ifndef FPNG_ORACLE_H
#define FPNG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPO_ENCODE_SLOWER 1u      /* src/fpng.h:34-38  */
#define FPO_FORCE_UNCOMPRESSED 2u /* src/fpng.h:40-41  */

/* src/fpng.cpp:393-401 (semantics of fpng_crc32: init 0 convention, ~ in/out inside). */
uint32_t fpo_crc32(const void *data, size_t size, uint32_t prev_crc);

/* src/fpng.cpp:480-487 (semantics of fpng_adler32, init 1). */
uint32_t fpo_adler32(const void *data, size_t size, uint32_t adler);

/* Upper bound of any outcome of fpo_encode = the raw-fallback size (src/fpng.cpp:1747). */
size_t fpo_max_encoded_size(uint32_t w, uint32_t h, uint32_t num_chans);

/*
 * src/fpng.cpp:1662-1803 fpng_encode_image_to_memory.  Returns 1 on success, 0 on bad
 * arguments (same conditions as the reference) or if out_cap is too small.
 */
int fpo_encode(const void *image, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t flags,
               uint8_t *out, size_t out_cap, size_t *out_size);

/*
 * Row-band form used by the multi-GPU stitching tests: tokenise rows [y0,y1) of the image into a
 * private bit string (starting at bit 0) with the 1-pass table for num_chans.  Returns the number
 * of bits; bits are LSB-first in `out` (which must be zeroed, capacity out_cap bytes).
 * Also returns the Adler partial sums (S1,S2 raw sums mod 65521, and length) of the band's
 * filtered bytes.  (src/fpng.cpp:1468-1558 / :1182-1241 restricted to a row range.)
 */
uint64_t fpo_encode_band_1pass(const void *image, uint32_t w, uint32_t h, uint32_t num_chans,
                               uint32_t y0, uint32_t y1, uint8_t *out, size_t out_cap,
                               uint32_t *adler_s1, uint32_t *adler_s2, uint64_t *adler_len,
                               uint32_t *last_unit_bits);

/* Row bands under either table (tests of the multi-GPU orchestration): see fpng_oracle.c */
void fpo_band_hist(const void *image, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t y0, uint32_t y1, uint32_t hist[288]);
uint64_t fpo_encode_band(const void *image, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t y0, uint32_t y1,
                         const uint32_t *lit_freq, uint8_t *out, size_t out_cap, uint32_t *adler_s1, uint32_t *adler_s2,
                         uint64_t *adler_len, uint32_t *last_unit_bits, uint32_t *first_token_bit, uint32_t *eob_bits,
                         uint32_t *eob_code, uint8_t *hdr);

/* Exposed for unit tests of the derived format tables. */
void fpo_get_1pass_table(uint32_t num_chans, uint8_t len_out[288], uint16_t code_out[288],
                         const uint8_t **prefix, uint32_t *prefix_len, uint32_t *start_bit);
void fpo_get_len_tables(uint16_t len_sym[256], uint8_t len_extra[256]);

/*
 * 2-pass table construction exposed for tests and for validating the device-side builder:
 * histogram (288 x u32) -> code lengths, codes and the dynamic block header bits.
 * hdr must hold >= 400 bytes; returns header length in bits (starting with the two zlib bytes,
 * BFINAL, BTYPE ...), i.e. the bit position where token bits start.
 * (src/fpng.cpp:868-907, :676-709, :746-816)
 */
uint32_t fpo_build_dynamic_table(const uint32_t lit_freq[288], uint32_t num_chans,
                                 uint8_t len_out[288], uint16_t code_out[288], uint8_t *hdr);

/*
 * Table training (src/fpng_test.cpp:766-973 + src/fpng.cpp:909-988, FPNG_TRAIN_HUFFMAN_TABLES builds of the reference): from a
 * corpus of images with num_chans channels to the code lengths / codes / block prefix of a new 1-pass table.  Same outputs as
 * fpo_build_dynamic_table.
 */
uint32_t fpo_train_tables(const void *const *images, const uint32_t *w, const uint32_t *h, uint32_t n, uint32_t num_chans,
                          uint8_t len_out[288], uint16_t code_out[288], uint8_t *hdr);

#ifdef __cplusplus
}
#endif
#endif
