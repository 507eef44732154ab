// encoder.h -- internals shared by the translation units of libfpng_amd.so (api.cpp: the C ABI of the device-resident path;
// pipeline.cpp: host-buffer pipelines, band planning, whole-node batches; sharded.cpp: one image over several GPUs).
#pragma once
#include "fpng_amd.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace fpng_amd {

// sets fpng_amd_last_error() of the calling thread and returns `code`
int fail(int code, const char *what, hipError_t e = hipSuccess);

#define HIP_TRY(expr)                                                                \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) return ::fpng_amd::fail(FPNG_AMD_ERR_HIP, #expr, e_);  \
    } while (0)

// Large device buffers are never handed back to the runtime while the process lives: device memory that was hipFree'd and
// comes back from a later hipMalloc downloads at 23 GB/s instead of 54 through the copy engines (measured: a second encoder
// made after the first one was destroyed took 3.7-4.0 ms per streamed 8K frame instead of 2.8; with the frees left out, 2.8 --
// profiles/r03_host_path.txt).  Blocks of 1 MiB and more go to a per-device list instead (api.cpp) and serve later requests
// they fit; fpng_amd_release_cached_memory() empties it.  `give` waits for the device first, as hipFree would.
void *device_take(size_t bytes, size_t *got);
void device_give(void *p, size_t bytes);

template <typename T> struct DeviceBuf {
    T *p = nullptr;
    size_t cap = 0;
    bool fresh = false; // set when ensure() (re)allocated: the contents are undefined
    // (giving a buffer up waits for the device: growing a buffer that a submission in flight still uses is safe, merely a
    // stall; capacities grow geometrically so that it stops happening after the first few submissions)
    int ensure(size_t n)
    {
        if (n <= cap) return FPNG_AMD_OK;
        const size_t want = std::max(n, cap + cap / 2);
        release();
        size_t got = 0;
        void *q = device_take(want * sizeof(T), &got);
        if (!q) {
            hipError_t e = hipMalloc(&q, want * sizeof(T));
            if (e != hipSuccess) return ::fpng_amd::fail(FPNG_AMD_ERR_OUT_OF_MEMORY, "hipMalloc scratch", e);
            got = want * sizeof(T);
        }
        p = (T *)q;
        cap = got / sizeof(T);
        fresh = true;
        return FPNG_AMD_OK;
    }
    void release()
    {
        if (p) device_give(p, cap * sizeof(T));
        p = nullptr;
        cap = 0;
    }
};
template <typename T> struct PinnedBuf {
    T *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n)
    {
        if (n <= cap) return FPNG_AMD_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        n = std::max(n, (size_t)16);
        hipError_t e = hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return ::fpng_amd::fail(FPNG_AMD_ERR_OUT_OF_MEMORY, "hipHostMalloc", e);
        cap = n;
        return FPNG_AMD_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};


// reference src/fpng.cpp:1670-1680 plus the 32-bit arithmetic limit of :1682-1705
int check_dims(uint32_t w, uint32_t h, uint32_t c);
// the 58 bytes in front of the zlib stream (reference src/fpng.cpp:1767-1791); the IDAT length (bytes 50..53) is left zero
void make_png_header(uint8_t *hdr60, uint32_t w, uint32_t h, uint32_t c);
// first token bit / end-of-block length of the 1-pass table (host copy of the format tables)
const TokenTable *host_1pass_table(uint32_t num_chans);

} // namespace fpng_amd

using namespace fpng_amd;

// A stream for one of the host paths' two copy directions.  The runtime serves all HIP streams of one priority from a pool of four
// hardware queues (its log: "maximum per priority is: 4"; an encoder alone makes five ordinary streams), and a copy it cannot
// give to an SDMA engine becomes a blit kernel on its stream's queue; streams of another priority come from another pool, so
// the copy streams ask for the highest one and never share a queue with the lanes' kernels.  (A precaution: with the two
// directions on two engines -- ensure_copy_streams() in pipeline.cpp -- we measured no difference.)
inline hipError_t create_copy_stream(hipStream_t *s)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || greatest == least)
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
}

struct fpng_amd_encoder {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    hipEvent_t ev[FPNG_AMD_NUM_PHASES + 1] = {};
    bool ev_ready = false;
    float phase_ms[FPNG_AMD_NUM_PHASES] = {};
    uint32_t phases_recorded = 0;

    PinnedBuf<Job> h_jobs;
    PinnedBuf<JobState> h_states;
    // Device scratch of one submission.  Batch submissions alternate between kLanes internal streams, each
    // with its own scratch, so that the tail of one submission (stored fallback, CRC, trailer) overlaps the
    // VALU-bound encode kernel of the next one.  Set 0 also serves the synchronous entry points (bands, wrap_png),
    // which drain the lanes first.
    struct Scratch {
        DeviceBuf<Job> d_jobs;
        DeviceBuf<RowInfo> d_rows;
        DeviceBuf<uint64_t> d_row_off;
        DeviceBuf<JobState> d_states;
        DeviceBuf<Result> d_results;
        DeviceBuf<uint32_t> d_partials;
        DeviceBuf<uint32_t> d_hist;
        size_t hist_zero = 0; // leading counters of d_hist that are zero when the lane's stream gets there (the table builder re-zeroes what it read)
        DeviceBuf<TokenTable> d_dyn;
        DeviceBuf<uint32_t> d_local; // rows pipeline: the rows' local streams (Job::local_base / local_stride)
        hipEvent_t last_done = nullptr; // `done` event (owned by a slot) of the last submission that used this set
        void release()
        {
            d_jobs.release(), d_rows.release(), d_row_off.release(), d_states.release(), d_results.release();
            d_partials.release(), d_hist.release(), d_dyn.release(), d_local.release();
        }
    };
    static constexpr int kLanes = 8; // most lanes an encoder can have; n_lanes of them exist and take submissions in turn
    uint32_t n_lanes = 2;            // FPNG_AMD_LANES, or by the hardware queues the process got (api.cpp, default_lanes()), read when the encoder is made
    Scratch sc[kLanes];
    hipStream_t lane_stream[kLanes] = {};
    hipEvent_t prev_walked = nullptr; // `walked` event of the previous submission (owned by its slot)
    DeviceBuf<uint8_t> d_stage_in, d_stage_out; // fpng_amd_encode_host
    // fpng_amd_encode_host_batch: ring of device staging buffers, one copy stream per direction
    struct HostRing {
        static constexpr int kDepth = 7; // (3 of them for big frames, all for small ones: fpng_amd_encode_host_batch)
        DeviceBuf<uint8_t> d_in[kDepth], d_out[kDepth];
        hipStream_t up = nullptr, down = nullptr;
    } host;
    // Submissions are pipelined: each one owns a slot of pinned host memory (job records going down, result
    // records coming back) guarded by an event, so fpng_amd_encode_submit() never waits for the GPU unless all
    // slots are in flight.  A submission's ticket is its sequence number; its records stay readable until its
    // slot is reused, kSlots submissions later.
    static constexpr int kSlots = 8;
    struct Slot {
        PinnedBuf<Job> jobs, jobs2; // jobs2: the second upload of 2-pass (tables patched)
        PinnedBuf<Result> results;
        hipEvent_t in = nullptr;     // recorded on the caller's stream: the inputs are ready
        hipEvent_t walked = nullptr; // recorded after the row walk
        hipEvent_t done = nullptr;   // recorded on the lane: PNGs and result records are complete
        bool in_flight = false;
        uint32_t n = 0;
        uint64_t ticket = 0;
    } slots[kSlots];
    uint64_t submitted = 0; // tickets handed out so far
    uint64_t band_token_bits = 0; // row bands: what fpng_amd_band_encode() left for fpng_amd_band_place()
    uint32_t band_eob_bits = 0;
    bool band_two_pass = false;
    bool last_two_pass = false;   // phase names of the last submission
    uint32_t band_crc_ranges = 0; // fpng_amd_band_place(): number of 64 KiB CRC ranges of the image
    uint64_t band_self_end = 0;   // placement with zlib_size == 0: file offset of the window's end (the CRC ranges hang off it)
    PinnedBuf<uint32_t> h_partials; // fpng_amd_band_crc(): the band's partials on their way to the host-side fold
    DeviceBuf<uint32_t> d_stream_partials; // fpng_amd_encode_host_to(): partials of all bands of the frame
    uint32_t last_host_bands = 0;  // row bands of the last fpng_amd_encode_host*() call (1 = the serial path)
    DeviceBuf<uint8_t> d_decode;  // fpng_amd_decode_batch(): all of its device scratch
    DeviceBuf<unsigned long long> d_dec_gran; // ... the look-back granules of dec_unfilter_kernel: zeroed when allocated, then told apart by epochs
    uint32_t dec_epoch = 0;
    // the decoder's lookup tables of the last batch that had at most kDecLutCache distinct ones (1-pass files: always the same two):
    // a batch whose tables are all here neither uploads code lengths nor builds tables (dec_build_lut_kernel: 34 us in front of everything)
    static constexpr uint32_t kDecLutCache = 4;
    DeviceBuf<uint32_t> d_lut_cache;
    uint8_t lut_cache_keys[kDecLutCache][288] = {};
    uint32_t lut_cache_n = 0;
    hipEvent_t dec_prof_ev[5] = {}; // profiling: around the decode kernels of the last call's first group of files
    bool dec_prof_recorded = false;
    hipStream_t dec_up = nullptr; // ... the stream the files' bytes are uploaded on, one event per group of files
    hipEvent_t dec_ev[16] = {};
    hipEvent_t dec_ev2[16] = {}, dec_ev3[16] = {}; // the streamed fpng_amd_decode_host: a piece's byte count is known / its finished rows are pixels
    PinnedBuf<uint8_t> h_dec_fetch; // fpng_amd_decode_batch_device(): the files' first and last bytes on their way to the host parser
    fpng_amd_sharded_report sharded_report = {}; // fpng_amd_encode_image_sharded(): where the last call's bytes went
    DeviceBuf<uint8_t> d_xchg;    // fpng_amd_encode_image_sharded(): the records it exchanges, and their pinned mirror
    PinnedBuf<uint8_t> h_xchg;
    struct HostWorkers *workers = nullptr; // fpng_amd_encode_host_to(): the uploader and downloader threads (pipeline.cpp)
    hipEvent_t band_copied[4] = {}; // the pinned job record h_jobs[k] of an asynchronous band call has been uploaded
};


namespace fpng_amd {
// Host-side wait for every submission in flight on the lanes.
int drain(fpng_amd_encoder *e);
// stops and joins the encoder's copy threads (no-op when there are none)
void destroy_host_workers(fpng_amd_encoder *e);
// has this host range been copied one direction at a time before (or page-locked through fpng_amd_pin_host_memory)?
int ensure_copy_streams(fpng_amd_encoder *e); // pipeline.cpp: the two copy streams of the host paths, on two copy engines
} // namespace fpng_amd
