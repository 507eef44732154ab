/*
 * fpng_amd.h -- C ABI of the MI355X-native fpng encode hot path (libfpng_amd.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types, no exceptions.
 * The reference has no FFI of its own -- its operator boundary is the nine free functions of
 * `namespace fpng` (reference src/fpng.h:17-111).  Each entry point below names the reference
 * interface it replaces or the part of it that it carries; the `fpng::` C++ drop-in
 * (include/fpng.h + fpng_amd/csrc/fpng_dropin.cpp) is a thin wrapper over exactly these calls, and
 * INTEGRATION.md shows the binding a maintainer of the reference (or of its Python/other
 * bindings) would add.
 *
 * All encode entry points produce output that is BYTE-IDENTICAL to the reference's
 * fpng_encode_image_to_memory() (reference src/fpng.cpp:1662-1803) for the same pixels and flags.
 *
 * Return convention: 0 = success, negative = FPNG_AMD_ERR_*.  fpng_amd_last_error() gives text.
 * There is NO CPU fallback: without a usable HIP device every encode call fails with
 * FPNG_AMD_ERR_NO_DEVICE.
 */
#ifndef FPNG_AMD_H
#define FPNG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPNG_AMD_ABI_VERSION 5

/* ---- status codes ---- */
#define FPNG_AMD_OK 0
#define FPNG_AMD_ERR_INVALID_ARG (-1)      /* the conditions of reference src/fpng.cpp:1670-1680 */
#define FPNG_AMD_ERR_NO_DEVICE (-2)
#define FPNG_AMD_ERR_HIP (-3)
#define FPNG_AMD_ERR_BUFFER_TOO_SMALL (-4)
#define FPNG_AMD_ERR_OUT_OF_MEMORY (-5)
#define FPNG_AMD_ERR_UNSUPPORTED (-6)      /* > 4 GiB of filtered bytes: undefined in the reference (src/fpng.cpp:1682-1705) */
#define FPNG_AMD_ERR_IO (-7)               /* a file could not be written (reference src/fpng.cpp:1818-1827 returns false) */

/* ---- encode flags: same bit values as reference src/fpng.h:34-42 ---- */
#define FPNG_AMD_ENCODE_SLOWER 1u      /* 2-pass: per-image dynamic Huffman table */
#define FPNG_AMD_FORCE_UNCOMPRESSED 2u /* stored Deflate blocks only */

/* ---- how an image ended up encoded (informational; the reference decides this silently) ---- */
#define FPNG_AMD_MODE_COMPRESSED 0u /* one dynamic-Huffman block */
#define FPNG_AMD_MODE_STORED 1u     /* raw fallback, reference src/fpng.cpp:1728-1758 */

/* ---- library ---- */

/* Device buffers of 1 MiB and more that encoders give up (on destruction, or when they outgrow them) are kept by the library
 * and handed to later requests of the same device instead of going back to the runtime -- memory that was freed and
 * allocated again downloads at half speed through the copy engines.  This call frees everything on that list
 * (all devices); buffers in use by live encoders are not touched. */
int fpng_amd_release_cached_memory(void);

/* Replaces fpng_init() (reference src/fpng.h:17): there it probes CPUID for SSE4.1, here it binds
 * the calling thread's default context to HIP device `device` (-1 = current device) and uploads
 * the format tables.  Optional, like the original: every other call self-initialises lazily and
 * outputs never depend on whether it ran. */
int fpng_amd_init(int device);

/* Replaces fpng_cpu_supports_sse41() (reference src/fpng.h:23): 1 if a gfx950-capable HIP device
 * is usable by this library, else 0. */
int fpng_amd_device_available(void);
int fpng_amd_device_count(void);
const char *fpng_amd_last_error(void);
int fpng_amd_abi_version(void);

/* What the library got from the HIP runtime, and what it does with it.  SIDE EFFECT OF LOADING THIS LIBRARY: its constructor sets
 * GPU_MAX_HW_QUEUES=8 in the process environment (the runtime spreads a process's streams over that many hardware queues, four by
 * default; with four the encoder's chains share queues and run one behind the other) -- unless the variable is set already,
 * FPNG_AMD_KEEP_HW_QUEUES=1 says hands off, or the process has /dev/kfd open (the runtime has read its settings: too late).  The
 * setting counts only before the process's first HIP call, is inherited by child processes and applies to every HIP user of the
 * process.  hw_queues = what the library believes the runtime uses (no HIP call reports it); source says why:
 *   LIBRARY_SET  the constructor set 8 (loaded before the first HIP call) -> four lanes
 *   CALLER_SET   the variable was there: its value is taken
 *   DRIVER_OPEN  loaded after the process's first HIP call (e.g. after `import torch; torch.cuda.init()`): four queues, two lanes --
 *                every batch result is the same, 8 x 8K encode throughput reads 3-5 % lower, one-frame submissions 25 % lower
 *   HANDS_OFF    FPNG_AMD_KEEP_HW_QUEUES=1
 * lanes = the lanes an encoder created NOW would have (FPNG_AMD_LANES overrides: 1..8); an encoder keeps the count it was made with. */
#define FPNG_AMD_HWQ_LIBRARY_SET 0
#define FPNG_AMD_HWQ_CALLER_SET 1
#define FPNG_AMD_HWQ_DRIVER_OPEN 2
#define FPNG_AMD_HWQ_HANDS_OFF 3
typedef struct fpng_amd_encoder fpng_amd_encoder;
typedef struct fpng_amd_runtime {
    uint32_t hw_queues;
    uint32_t hw_queue_source; /* FPNG_AMD_HWQ_* */
    uint32_t lanes;
    uint32_t reserved[5];
} fpng_amd_runtime;
int fpng_amd_runtime_info(fpng_amd_runtime *info);
/* ... and the lanes of a live encoder */
uint32_t fpng_amd_encoder_lanes(const fpng_amd_encoder *enc);

/* fpng_crc32 / fpng_adler32 (reference src/fpng.h:26-31), host buffers, same calling convention
 * (prev = finished checksum of the preceding bytes; init 0 / 1). */
uint32_t fpng_amd_crc32(const void *data, size_t size, uint32_t prev_crc32);
uint32_t fpng_amd_adler32(const void *data, size_t size, uint32_t adler);
/* checksum of X||Y from the checksums of X and Y and |Y| -- what makes the path shardable */
uint32_t fpng_amd_crc32_combine(uint32_t crc_x, uint32_t crc_y, uint64_t len_y);
uint32_t fpng_amd_adler32_combine(uint32_t adler_x, uint32_t adler_y, uint64_t len_y);

/* Largest possible output of any encode call for these dimensions (= the stored-block size,
 * reference src/fpng.cpp:1747, + container).  d_out capacities must be >= this. */
size_t fpng_amd_max_encoded_size(uint32_t w, uint32_t h, uint32_t num_chans);

/* ---- environment knobs (read once per process; the defaults are the measured optimum; INTEGRATION.md section 8 has the table):
 *      FPNG_AMD_KEEP_HW_QUEUES=1    the library's constructor leaves GPU_MAX_HW_QUEUES alone (fpng_amd_runtime_info above)
 *      FPNG_AMD_LANES=1..8          internal streams that take submissions in turn (default: 4 over eight hardware queues, else 2;
 *                                   1 serialises everything); read when an encoder is created
 *      FPNG_AMD_LOCAL_LIMIT_MB=n    cap on the scratch for the rows' local streams (default 98304); a submission that would
 *                                   need more is refused with FPNG_AMD_ERR_OUT_OF_MEMORY before anything is launched
 *      FPNG_AMD_STAGGER=0|1         2-pass: make a submission's row walk wait for the previous submission's walk
 *                                   (default: on for FPNG_AMD_ENCODE_SLOWER with two lanes, off otherwise)
 *      FPNG_AMD_HOST_BANDS=n        fpng_amd_encode_host(): row bands of the streamed upload/encode/download pipeline for every
 *                                   1-pass frame (default: by image size, for page-locked or previously seen buffers; 1 = serial)
 *      FPNG_AMD_DECODE_CPU=1        fpng::fpng_decode_memory never uses the GPU decoder
 *      FPNG_AMD_DECODE_MAX_ROUNDS=n synchronisation rounds before a file is reported FPNG_AMD_DECODE_UNDECIDED (default 64)
 *      FPNG_AMD_DECODE_STREAM=0     fpng_amd_decode_host(): upload, decode, download one after the other
 *      FPNG_AMD_TRACE=1             timelines of the streamed host paths, the sharded path and the decoder on stderr ---- */

/* ---- encoder object: the caller's HIP stream (ordering point) + two internal streams ("lanes") with
 *      reusable device scratch.  Not thread-safe; create one per thread (the reference is re-entrant,
 *      reference src/fpng.cpp:371 note in SURVEY 8b). ---- */
typedef struct fpng_amd_encoder fpng_amd_encoder;

/* hip_stream: the hipStream_t submissions are ordered against (e.g. torch's current stream: the producer
 * of the pixels), or NULL to let the encoder create its own non-blocking stream. */
int fpng_amd_encoder_create(fpng_amd_encoder **enc, int device, void *hip_stream);
/* The same, but hip_stream is taken literally: NULL means the legacy default (null) stream -- e.g. torch's default
 * stream -- rather than "create one".  Submissions are then ordered behind everything already enqueued there. */
int fpng_amd_encoder_create_on_stream(fpng_amd_encoder **enc, int device, void *hip_stream);
/* Re-point the ordering stream (e.g. to torch's current stream before a submission); NULL = the null stream. */
int fpng_amd_encoder_set_stream(fpng_amd_encoder *enc, void *hip_stream);
void fpng_amd_encoder_destroy(fpng_amd_encoder *enc);
void *fpng_amd_encoder_stream(fpng_amd_encoder *enc);

typedef struct fpng_amd_image {
    const void *d_pixels; /* DEVICE pointer, R first, pitch = w*num_chans (reference src/fpng.h:44-47) */
    uint32_t w, h, num_chans;
    uint8_t *d_out;  /* DEVICE pointer, receives the whole .png file */
    size_t out_cap;  /* >= fpng_amd_max_encoded_size(w,h,num_chans) */
} fpng_amd_image;

typedef struct fpng_amd_result {
    uint64_t png_size; /* bytes written to d_out */
    uint32_t mode;     /* FPNG_AMD_MODE_* */
    uint32_t status;   /* 0 = ok */
} fpng_amd_result;

/*
 * THE HOT PATH.  fpng_encode_image_to_memory() (reference src/fpng.h:48, src/fpng.cpp:1662-1803)
 * for `n` device-resident images in one submission.  Returns without waiting; nothing is copied to the
 * host except the n result records, which are delivered by fpng_amd_encode_finish().
 *
 * Ordering: the kernels start after everything enqueued on the encoder's stream before this call, and run
 * on one of the encoder's two internal lanes, so consecutive submissions OVERLAP on the GPU (the serial
 * tail and the LDS-bound CRC of one under the row walkers of the next).  Outputs are complete after
 * fpng_amd_encode_finish() (host side) or, for work enqueued on the encoder's stream, after
 * fpng_amd_encoder_join().  The pixel and output buffers of submissions in flight must not alias.
 */
int fpng_amd_encode_batch_async(fpng_amd_encoder *enc, const fpng_amd_image *images, uint32_t n, uint32_t flags);

/* The same call handing back a TICKET for this submission, so that a pipeline with several submissions in flight can
 * collect each one's result records: fpng_amd_encode_wait() waits for that submission only (later ones keep
 * running) and copies out its n records; fpng_amd_encode_query() polls (1 done, 0 running).  Records stay
 * retrievable until 8 further submissions have been made.  A failed submit hands out no ticket and leaves the encoder as it
 * was (the ticket is committed after the last call that can fail). */
int fpng_amd_encode_submit(fpng_amd_encoder *enc, const fpng_amd_image *images, uint32_t n, uint32_t flags, uint64_t *ticket);
int fpng_amd_encode_wait(fpng_amd_encoder *enc, uint64_t ticket, fpng_amd_result *results, uint32_t n);
int fpng_amd_encode_query(fpng_amd_encoder *enc, uint64_t ticket);

/* Device-side join: the encoder's stream waits (no host wait) for every submission made so far. */
int fpng_amd_encoder_join(fpng_amd_encoder *enc);

/* Waits for ALL outstanding fpng_amd_encode_batch_async() submissions of this encoder and copies out the
 * n result records of the last one (results may be NULL to just wait). */
int fpng_amd_encode_finish(fpng_amd_encoder *enc, fpng_amd_result *results, uint32_t n);

/* Convenience: fpng_encode_image_to_memory() on HOST buffers (H2D + encode + D2H, synchronous).
 * This is what the fpng:: C++ drop-in calls. */
int fpng_amd_encode_host(fpng_amd_encoder *enc, const void *pixels, uint32_t w, uint32_t h, uint32_t num_chans,
                         uint32_t flags, uint8_t *out, size_t out_cap, size_t *out_size);

/* The same with the OUTPUT ALLOCATOR in the caller's hands -- what a std::vector-based caller (the fpng:: drop-in) needs so
 * that nothing is zero-filled or copied twice: `reserve(user, bytes)` must return a buffer of at least `bytes` bytes that
 * still holds everything written so far (NULL = give up: FPNG_AMD_ERR_BUFFER_TOO_SMALL); it is called with growing sizes, only
 * while the call runs and never concurrently, possibly from a helper thread.  Large 1-pass frames are STREAMED: the frame is
 * cut into row bands, band k+1 is uploaded while band k is encoded and placed and band k-1's piece of the file is downloaded,
 * so a call takes about max(upload, download) instead of their sum (one 8K RGBA frame: 2.8 instead of 3.6 ms).
 * Which frames are streamed: 1-pass frames of 16 MiB of pixels and more, in pageable or page-locked memory alike.
 * fpng_amd_pin_host_memory() (= hipHostRegister; undo with fpng_amd_unpin_host_memory() BEFORE freeing the buffer) saves the
 * runtime's pinning of each chunk it copies (8K RGBA: 2.80 instead of 2.85 ms), nothing more. */
typedef uint8_t *(*fpng_amd_reserve_fn)(void *user, size_t bytes);
int fpng_amd_encode_host_to(fpng_amd_encoder *enc, const void *pixels, uint32_t w, uint32_t h, uint32_t num_chans,
                            uint32_t flags, fpng_amd_reserve_fn reserve, void *user, size_t *out_size);
/* Row bands the last fpng_amd_encode_host() / _host_to() call of this encoder was streamed in (1 = the serial path). */
int fpng_amd_encoder_last_host_bands(fpng_amd_encoder *enc);
int fpng_amd_pin_host_memory(void *p, size_t bytes);
int fpng_amd_unpin_host_memory(void *p);

/* fpng_encode_image_to_memory() / fpng_encode_image_to_file() (reference src/fpng.h:48-52, src/fpng.cpp:1806-1828) for MANY
 * frames in host memory: uploads, encodes, downloads and file writes of consecutive frames overlap (a ring of device
 * staging buffers, an uploader and a downloader thread, n_writer_threads file writers; 0 = the downloader writes).
 * Per frame: `out` (caller buffer of out_cap bytes, may be NULL when `path` is set), `path` (may be NULL), *out_size. */
typedef struct fpng_amd_host_image {
    const void *pixels;
    uint32_t w, h, num_chans, reserved;
    uint8_t *out;
    size_t out_cap;
    size_t *out_size;
    const char *path;
} fpng_amd_host_image;
int fpng_amd_encode_host_batch(fpng_amd_encoder *enc, const fpng_amd_host_image *images, uint32_t n, uint32_t flags,
                               int n_writer_threads);

/* ---- whole node from ONE process (SURVEY 8b "a multi-GPU form"): one encoder + staging ring per listed device, the frames
 *      of a host batch are dealt round-robin to the devices, every device runs fpng_amd_encode_host_batch() on its share from
 *      its own thread.  `devices` may name a device more than once (two pipelines on one GPU). ---- */
typedef struct fpng_amd_node fpng_amd_node;
int fpng_amd_node_create(fpng_amd_node **node, const int *devices, uint32_t n_devices);
void fpng_amd_node_destroy(fpng_amd_node *node);
uint32_t fpng_amd_node_size(const fpng_amd_node *node);
int fpng_amd_node_encode_host_batch(fpng_amd_node *node, const fpng_amd_host_image *images, uint32_t n, uint32_t flags,
                                    int n_writer_threads_per_device);
/* ONE host-resident image over the node's devices (reference src/fpng.cpp:1662-1803 writes the same
 * file): contiguous row bands, one per listed device; every device uploads ITS band and downloads ITS window of the file
 * over its own PCIe link, straight from / into the caller's memory (a 16384 x 16384 RGBA image: 1.07 GB up + 0.47 GB down --
 * 29 ms on one link, under 4 ms on eight).  The bands' 64-byte records (and, 2-pass, their histograms) meet on the host; no
 * device-to-device traffic.  `reserve` as in fpng_amd_encode_host_to(): called once, with the file's exact size. */
int fpng_amd_node_encode_host_image(fpng_amd_node *node, const void *pixels, uint32_t w, uint32_t h, uint32_t num_chans, uint32_t flags,
                                    fpng_amd_reserve_fn reserve, void *user, size_t *out_size);

/* ---- row-band interface: one image sharded by rows over several GPUs.
 *      The stream stays ONE IDAT / ONE Deflate block; bands are stitched at bit granularity. ---- */

typedef struct fpng_amd_band {
    const void *d_rows;      /* DEVICE pointer to row y0 of the image (pitch w*num_chans) */
    const void *d_row_above; /* row y0-1 (ignored when y0 == 0): the Up filter's only dependency across bands */
    uint32_t w, num_chans;
    uint32_t y0, y1;         /* rows [y0, y1) of the image */
    uint32_t h_total;        /* height of the whole image */
    uint32_t reserved;
} fpng_amd_band;

typedef struct fpng_amd_band_stats {
    uint64_t token_bits;  /* bits of all tokens of the band's rows */
    uint32_t adler_s1;    /* raw byte sum of the band's filtered bytes mod 65521 */
    uint32_t adler_s2;    /* raw position-weighted sum mod 65521 */
    uint64_t adler_len;   /* filtered bytes in the band = (w*c+1)*rows */
    uint32_t last_unit_bits;  /* size of the band's final flush unit (failure rule) */
    uint32_t first_token_bit; /* where row tokens start in the zlib stream: 490 / 503 for 1-pass (reference src/fpng.cpp:535,
                                 :551), the dynamic header's length for 2-pass -- the same on every rank */
    uint32_t eob_bits;        /* length of the end-of-block code of the table in use */
    uint32_t reserved;
} fpng_amd_band_stats;

/* 2-pass only (FPNG_AMD_ENCODE_SLOWER), asynchronous on the encoder's stream: d_hist288[0..288) (device, uint32) = symbol
 * histogram of the band (reference src/fpng.cpp:1021-1084 / :1299-1363).  The caller sums the bands' histograms over the
 * ranks (one all-reduce of 1152 bytes) and hands the result to fpng_amd_band_encode(): every rank builds the same table. */
int fpng_amd_band_hist(fpng_amd_encoder *enc, const fpng_amd_band *band, uint32_t *d_hist288);

/* Phase 1: the band's rows are encoded into the encoder's scratch streams (the same encode_rows kernel as whole images);
 * the band's counts come back to the host -- the only synchronisation of the band path: `stats` is what the ranks
 * exchange (one all_gather of a few words).  flags: 0 or FPNG_AMD_ENCODE_SLOWER (then d_hist288 = the image's histogram). */
int fpng_amd_band_encode(fpng_amd_encoder *enc, const fpng_amd_band *band, uint32_t flags, const uint32_t *d_hist288,
                         fpng_amd_band_stats *stats);

/* Phase 2, asynchronous on the encoder's stream: the streams of the last fpng_amd_band_encode() are shifted to stream
 * bit `start_bit` (first band: stats.first_token_bit) of an image whose zlib stream has `zlib_size` bytes, into a
 * WINDOW of whole 16-byte pieces of the file: d_window[0] is file byte *window_file_offset (a multiple of 16; 0 for the
 * first band, whose window also receives the stream's head), *window_bytes bytes are defined; bits that belong to other
 * bands are 0, so neighbouring windows are merged by OR-ing their one shared 16-byte piece.  The image's last band
 * (y1 == h_total) appends the end-of-block code.  PNG header, Adler-32, CRC and IEND are fpng_amd_wrap_png()'s.
 * zlib_size == 0: the size of the stream is not known yet (bands placed one after another while later ones are still being
 * uploaded or encoded): the CRC ranges are then laid out from the window's own end and fpng_amd_band_crc() turns them into
 * ONE raw CRC value per band, which fpng_amd_idat_crc_from_bands() combines once the last band is in. */
int fpng_amd_band_place(fpng_amd_encoder *enc, const fpng_amd_band *band, uint64_t start_bit, uint64_t zlib_size,
                        uint8_t *d_window, size_t window_cap, uint64_t *window_file_offset, size_t *window_bytes);

/* Wrap an assembled zlib stream (device memory, at d_png + 58, its last 4 bytes = room for the Adler-32) into the PNG
 * container: 58-byte header, big-endian `adler`, IDAT CRC-32, IEND (reference src/fpng.cpp:1764-1800).  Asynchronous on
 * the encoder's stream; *png_size = 58 + zlib_size + 16 is returned at once. */
int fpng_amd_wrap_png(fpng_amd_encoder *enc, uint8_t *d_png, size_t zlib_size, uint32_t adler, uint32_t w, uint32_t h,
                      uint32_t num_chans, size_t *png_size);

/* CRC-32 of the IDAT sharded like the rows: a raw CRC is linear, and a band's window is zero wherever other bands' bits
 * are, so the per-64-KiB-range CRC partials that fpng_amd_band_place() computes while it writes the window XOR together
 * across the bands to the partials of the whole file.  fpng_amd_band_crc_partials() copies this band's *n_partials
 * values (the same count on every rank: it depends on zlib_size only; d_partials = NULL just reports it) to device memory,
 * asynchronously on the encoder's stream; the caller XORs the ranks' arrays element-wise (a gather of a few KiB) and hands the result to
 * fpng_amd_wrap_png_crc(), which then finishes the file without reading it (fpng_amd_wrap_png() runs a CRC pass over
 * the whole file instead). */
int fpng_amd_band_crc_partials(fpng_amd_encoder *enc, uint32_t *d_partials, uint32_t cap, uint32_t *n_partials);
int fpng_amd_wrap_png_crc(fpng_amd_encoder *enc, uint8_t *d_png, size_t zlib_size, uint32_t adler, uint32_t w, uint32_t h,
                          uint32_t num_chans, const uint32_t *d_crc_partials, uint32_t n_partials, size_t *png_size);

/* The raw CRC-32 (init 0, no final xor) of the window of the band placed last with zlib_size == 0, foreign bits counting as
 * zero, as if the file ended at *end_offset (the window's end).  Waits for the placement. */
int fpng_amd_band_crc(fpng_amd_encoder *enc, uint32_t *raw_crc, uint64_t *end_offset);

/* ---- host-side arithmetic of the band path (no GPU needed): what every rank -- or one host streaming bands through one GPU
 *      -- derives from the bands' records.  Bands in row order; a band without rows has adler_len == 0. ---- */
typedef struct fpng_amd_band_plan {
    uint64_t end_bit;   /* zlib bit position after the last row token */
    uint64_t zlib_size; /* bytes of the compressed zlib stream incl. the Adler-32 */
    uint32_t adler;     /* Adler-32 of the image's filtered bytes */
    uint32_t stored;    /* 1: the reference's coder would have run out of buffer (src/fpng.cpp:567-588): encode the image whole,
                           it becomes stored blocks */
} fpng_amd_band_plan;
/* start_bits[k] = zlib bit where band k's tokens begin (exclusive prefix sum from stats[0].first_token_bit); flags: 0 or
 * FPNG_AMD_ENCODE_SLOWER (the failure rule differs, reference src/fpng.cpp:1169 / :1455). */
int fpng_amd_plan_bands(const fpng_amd_band_stats *stats, uint32_t n_bands, uint32_t w, uint32_t h, uint32_t num_chans,
                        uint32_t flags, uint64_t *start_bits, fpng_amd_band_plan *plan);
/* The window fpng_amd_band_place() writes for a band: whole 16-byte pieces of the file; *shared_head = 16 when its first
 * piece is also the last piece of the previous band's window (OR the two), else 0. */
int fpng_amd_band_window(int is_first, int is_last, uint64_t start_bit, uint64_t token_bits, uint32_t eob_bits,
                         uint64_t *file_offset, size_t *bytes, uint32_t *shared_head);
/* IDAT chunk CRC-32 (reference src/fpng.cpp:1797-1800) from the bands' raw window CRCs (fpng_amd_band_crc). */
uint32_t fpng_amd_idat_crc_from_bands(const uint32_t *raw_crc, const uint64_t *end_offset, uint32_t n_bands, uint64_t zlib_size,
                                      uint32_t adler);
/* The 58 bytes in front of the zlib stream (signature, IHDR, fdEC, IDAT length + type; reference src/fpng.cpp:1767-1791) and
 * the 20 bytes behind its tokens (Adler-32 big-endian = the stream's last 4 bytes, IDAT CRC-32, IEND). */
int fpng_amd_png_head(uint32_t w, uint32_t h, uint32_t num_chans, uint64_t zlib_size, uint8_t head[58]);
void fpng_amd_png_tail(uint32_t adler, uint32_t idat_crc, uint8_t tail[20]);

/* ---- ONE image over several GPUs from C/C++ (SURVEY 8b "a multi-GPU form", 8e): the same steps as fpng_amd/sharded.py on
 *      top of a TRANSPORT the caller provides (MPI, a custom fabric ...) or the built-in RCCL one.  One process (or thread)
 *      per GPU, each with its own encoder and the rows [y0, y1) of the image; the finished .png appears in `d_png` on rank
 *      `root`.  Exchanges: (2-pass: one all-reduce of 288 counters,) one all-gather of a 64-byte record, one of a 16-byte
 *      record, and every non-root band's window (its share of the file) sent to the root -- no pixel leaves its GPU unless
 *      the image turns out incompressible (the reference's stored-block outcome: the rows are then collected on the root).
 *      All buffers are device pointers; every call is made on `stream` (the encoder's) ---- */
typedef struct fpng_amd_transport {
    void *ctx;
    int rank, world;
    int (*all_gather)(void *ctx, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
    int (*all_reduce_sum_u32)(void *ctx, void *d_buf, size_t count, void *stream); /* in place */
    int (*group_begin)(void *ctx); /* the sends / receives up to group_end() may complete in any order */
    int (*send)(void *ctx, const void *d_buf, size_t bytes, int peer, void *stream);
    int (*recv)(void *ctx, void *d_buf, size_t bytes, int peer, void *stream);
    int (*group_end)(void *ctx);
} fpng_amd_transport;

/* The built-in transport: RCCL (librccl.so.1 is loaded at run time, so libfpng_amd.so itself does not depend on it).  Rank 0
 * makes the 128-byte id, the caller gets it to the other ranks (a file, a socket, MPI_Bcast, torch.distributed ...), then
 * every rank creates its transport; `device` = the HIP device of this rank's encoder. */
int fpng_amd_rccl_unique_id(uint8_t id[128]);
int fpng_amd_rccl_transport_create(fpng_amd_transport **t, const uint8_t id[128], int rank, int world, int device);
void fpng_amd_rccl_transport_destroy(fpng_amd_transport *t);

/* band: this rank's rows (y1 == y0 allowed: a rank without rows still takes part in the collectives); flags: 0 or
 * FPNG_AMD_ENCODE_SLOWER; d_png / png_cap (>= fpng_amd_max_encoded_size() + 64) / png_size matter on the root only.  Returns
 * when the file is complete on the root (the other ranks return when their sends are enqueued and the stream is drained). */
int fpng_amd_encode_image_sharded(fpng_amd_encoder *enc, const fpng_amd_transport *t, const fpng_amd_band *band, uint32_t flags,
                                  int root, uint8_t *d_png, size_t png_cap, size_t *png_size);

/* Where the bytes of this rank's last fpng_amd_encode_image_sharded() call went (what a scaling run is held against: DESIGN.md 5):
 * every window travels ONCE, from its rank straight into its place in the root's file buffer -- there is no staging copy on the
 * root for a compressed image (root_staged_bytes stays 0; only the stored outcome, where the rows themselves go to the root, stages). */
typedef struct fpng_amd_sharded_report {
    uint64_t sent_bytes;          /* this rank -> root (its window, or its rows for the stored outcome) */
    uint64_t received_in_place;   /* root: bytes received straight into d_png at their file offset */
    uint64_t root_staged_bytes;   /* root: bytes that went through a staging buffer and were copied again (stored outcome: the rows) */
    uint64_t own_window_bytes;    /* this rank's band placed by its own kernels (root: directly in the file) */
    uint32_t shared_pieces;       /* root: 16-byte pieces two windows share, OR-ed in afterwards */
    uint32_t collectives;         /* all-gathers / all-reduces of the call (2 or 3) */
    uint32_t stored;              /* 1: the image ended as stored blocks */
    uint32_t reserved;
} fpng_amd_sharded_report;
int fpng_amd_sharded_last_report(fpng_amd_encoder *enc, fpng_amd_sharded_report *report);

/* First token bit of the 1-pass stream (490 for 4 channels, 503 for 3; reference src/fpng.cpp:535,:551)
 * and the EOB length (12) -- what a host needs to evaluate the failure rule without a GPU. */
int fpng_amd_1pass_layout(uint32_t num_chans, uint32_t *first_token_bit, uint32_t *eob_bits, uint32_t *prefix_bytes);

/* ---- GPU batch DECODE of fpng-written files (reference src/fpng.h:55-111 fpng_decode_memory; src/fpng.cpp:2209-2901) in its
 *      data-parallel form: container and block header are parsed on the host, the pixel stream is decoded by thousands of
 *      threads that find the token boundaries through the self-synchronisation of Huffman codes (decode.hip).
 *      fpng_amd_decode_batch: `data` is a HOST pointer to the whole .png (uploaded in groups while earlier groups decode);
 *      fpng_amd_decode_batch_device: `data` is a DEVICE pointer to the whole .png (e.g. what fpng_amd_encode_submit() wrote):
 *      only the file's first 1024 and last 64 bytes come back to the host for the container walk and the block header; the
 *      bytes must be complete when the call is made (or be produced on the encoder's stream).  Either way the pixels
 *      (desired_chans = 3 or 4 per pixel, like the reference's desired_channels) appear in DEVICE memory and the call returns
 *      when they are there.  status = the reference's fpng::FPNG_DECODE_* code for this file (0 = success);
 *      FPNG_AMD_DECODE_UNDECIDED: the GPU path leaves the file to the CPU decoder -- its token boundaries did not synchronise
 *      in the allotted rounds (FPNG_AMD_DECODE_MAX_ROUNDS in the environment, default 64; 0 = every compressed file), or it is
 *      a stored-block file cut into other block sizes than the encoder's (or with the one zero byte behind the image that the
 *      reference lets pass), a match at a row's first pixel, a table with codes for the reserved length symbols 286 / 287 (the
 *      reference's 4-channel decoder gives them a meaning) -- nothing an fpng encoder writes; decode it with
 *      fpng::fpng_decode_memory() instead; never reported for a file the CPU decoder would reject with a different code than
 *      FPNG_DECODE_NOT_FPNG. ---- */
#define FPNG_AMD_DECODE_UNDECIDED 64
typedef struct fpng_amd_png {
    const void *data; /* the file: HOST memory for fpng_amd_decode_batch, DEVICE memory for fpng_amd_decode_batch_device */
    uint32_t size;
    uint32_t reserved;
    uint8_t *d_pixels; /* DEVICE: receives w * h * desired_chans bytes */
    size_t pixels_cap;
} fpng_amd_png;
typedef struct fpng_amd_decode_result {
    uint32_t w, h, channels_in_file;
    int32_t status;
} fpng_amd_decode_result;
int fpng_amd_decode_batch(fpng_amd_encoder *enc, const fpng_amd_png *files, uint32_t n, uint32_t desired_chans,
                          fpng_amd_decode_result *results);
int fpng_amd_decode_batch_device(fpng_amd_encoder *enc, const fpng_amd_png *files, uint32_t n, uint32_t desired_chans,
                                 fpng_amd_decode_result *results);
/* One HOST-resident file to HOST pixels (reference src/fpng.h:108 fpng_decode_memory; the fpng:: drop-in routes images of
 * 256K pixels and more through it): container checks, upload, GPU decode, download into memory obtained from `reserve`.
 * `reserve` is called with w * h * desired_chans once the container and the block header are accepted -- BEFORE the stream is known
 * to decode when the file is streamed (8 MiB of IDAT and more: rows go down while later pieces come up), and possibly a second
 * time with the same size when such a file needs more synchronisation rounds; it must return the same or equally good memory
 * every time.  When result->status is not 0 the memory's contents are undefined (rows decoded before the damage may be there).
 * result->status as in fpng_amd_decode_batch(): FPNG_AMD_DECODE_UNDECIDED = decode it on the CPU. */
int fpng_amd_decode_host(fpng_amd_encoder *enc, const void *png, uint32_t size, uint32_t desired_chans, fpng_amd_reserve_fn reserve,
                         void *user, fpng_amd_decode_result *result);
/* What fpng_amd_decode_batch() settles on the HOST about one file before the GPU sees it (no GPU needed; tests/ run the decode
 * kernels' per-thread code on the CPU against it): result (container status, geometry; status 0 = the stream's shape is acceptable so
 * far), mode (0 = one dynamic block, 1 = stored blocks), the IDAT chunk's offset and payload length, the first row token's bit and
 * the bit no token may start at or behind (both counted from the zlib stream's first byte = png + idat_ofs + 8), and the kernels'
 * lookup table, FPNG_AMD_DECODE_LUT_WORDS words (fpng_amd/csrc/decode_core.h): lut[next 12 bits] = for a group of literals or a
 * match without extra bits: ALL the stream bits it takes << 28 | number of literals (1..3) << 26 | their byte values (first
 * lowest), or literals 0: bit 25 set, the match's length in bits 8..0; top four bits 0: a match with extra bits (bit 25, base
 * length in bits 8..0, extra bit count in bits 11..9, the length symbol's code bits in bits 15..12), the end of the block (bit 24,
 * code bits in bits 15..12), or 0 = no such code; then 64 words = the literals' code lengths. */
#define FPNG_AMD_DECODE_LUT_WORDS 4160
int fpng_amd_decode_plan(const void *png, uint32_t size, fpng_amd_decode_result *result, uint32_t *mode, uint32_t *idat_ofs,
                         uint32_t *idat_len, uint64_t *first_bit, uint64_t *end_limit_bit, uint32_t *lut);

/* ---- table training (reference src/fpng_test.cpp:766-973 "-t" + src/fpng.cpp:909-988, both only in builds of the reference with
 *      FPNG_TRAIN_HUFFMAN_TABLES=1): from a corpus of `n` device-resident images, all with num_chans channels (the reference's
 *      harness trains the 24 bpp and the 32 bpp table on the opaque and the translucent files separately), to a new 1-pass
 *      table in the form the reference prints it: the Deflate block prefix (whole bytes + `bit_buf_size` pending bits) and the
 *      288 code / code size pairs of g_dyn_huff_{3|4}_codes.  Histograms, their 16-bit adjustment and the table builder run
 *      on the device.  d_out / out_cap of the images are not used.  Note: a table trained here is a DIFFERENT FORMAT TABLE --
 *      files written with it are valid PNGs but not byte-identical to the stock fpng encoder's. ---- */
int fpng_amd_train_tables(fpng_amd_encoder *enc, const fpng_amd_image *images, uint32_t n, uint32_t num_chans, uint8_t *prefix,
                          size_t prefix_cap, size_t *prefix_bytes, uint32_t *bit_buf, uint32_t *bit_buf_size, uint32_t codes[288],
                          uint8_t code_sizes[288]);

/* ---- synthetic inputs for tests/bench, host memory ---- */
#define FPNG_AMD_SYNTH_NOISE 0
#define FPNG_AMD_SYNTH_SOLID 1
#define FPNG_AMD_SYNTH_GRAD 2
#define FPNG_AMD_SYNTH_BLOCKS 3
int fpng_amd_synth_image(int kind, uint32_t seed, uint32_t w, uint32_t h, uint32_t num_chans, uint8_t *out);

/* ---- instrumentation for bench.py: per-kernel durations of the last submission measured with HIP
 *      events (ms).  With profiling enabled submissions use one lane, i.e. they do not overlap.
 *      fpng_amd_encoder_phase_names(): comma-separated names of the phases of the last submission's
 *      launch chain: "encode_rows,scan,stored,assemble,finalize", with "hist,build_dynamic," in front for
 *      FPNG_AMD_ENCODE_SLOWER submissions. ---- */
#define FPNG_AMD_NUM_PHASES 8
int fpng_amd_encoder_set_profiling(fpng_amd_encoder *enc, int enabled);
int fpng_amd_encoder_last_phase_ms(fpng_amd_encoder *enc, float ms[FPNG_AMD_NUM_PHASES]);
const char *fpng_amd_encoder_phase_names(fpng_amd_encoder *enc);
/* ... and of the last fpng_amd_decode_batch*() call (first group of files) while profiling was enabled:
 *      "sync,offsets,emit,unfilter" = dec_sync_kernel (all rounds), dec_offsets_kernel + dec_subscan_kernel, nothing (the slot of
 *      round 5's dec_emit_kernel: since round 6 every token is decoded once, by the synchronisation; it reads a few microseconds),
 *      dec_unfilter_kernel + dec_stored_kernel (fpng_amd/csrc/decode.hip). */
#define FPNG_AMD_NUM_DECODE_PHASES 4
int fpng_amd_decode_last_phase_ms(fpng_amd_encoder *enc, float ms[FPNG_AMD_NUM_DECODE_PHASES]);

/* Instrumentation of the timing build (libfpng_amd_timing.so, -DFPNG_BUILD_TIMING): n_words = 8 reads the cycle counters
 * that build_dynamic_kernel left at the head of lane `lane`'s histogram scratch (dst[7] = 0xFEED selects the second page:
 * the phases inside the table builder). */
int fpng_amd_debug_peek(fpng_amd_encoder *enc, int lane, uint32_t *dst, uint32_t n_words);

#ifdef __cplusplus
}
#endif
#endif /* FPNG_AMD_H */
