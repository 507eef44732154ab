// api.cpp -- the C ABI of libfpng_amd.so (see include/fpng_amd.h) on top of the HIP kernels.
//
// Host-side orchestration only: argument rules of the reference's fpng_encode_image_to_memory
// (reference src/fpng.cpp:1662-1680), job descriptors, scratch management, kernel launches on one
// stream.  There is no CPU implementation of the encode path in this library.
#include "encoder.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>
#include <unistd.h>

namespace {
thread_local std::string g_last_error;

// HARDWARE QUEUES.  The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues, four by default: with the
// caller's stream, the encoder's stream and the lanes (one stream + scratch set each, whole chains alternate between them) two chains
// land in ONE queue and run one behind the other.  With eight queues and four lanes (same box, alternating; profiles/r05_hw_queues.txt)
// one 8K frame per submission + 29 %, 8 x 8K + 3 ... 4 % (2-pass + 6 %), 256 x 1080p RGB + 12 %, 1024 x 512^2 + 5 %; four lanes over
// four queues lose 7 % in 2-pass, so the lanes follow what the library can tell about the queues.  The runtime reads the variable
// when it initialises (the first HIP call of the process): the library asks for eight when it is loaded, unless the variable is set
// already, FPNG_AMD_KEEP_HW_QUEUES=1 says hands off, or the process has the GPU driver open already (too late: two lanes then).
int g_hw_queues = 4;
int g_hw_queue_source = FPNG_AMD_HWQ_DRIVER_OPEN; // fpng_amd_runtime_info(): why g_hw_queues is what it is

bool gpu_driver_open()
{
    DIR *d = opendir("/proc/self/fd");
    if (!d) return false;
    bool found = false;
    char path[64], target[256];
    while (const dirent *ent = readdir(d)) {
        if (ent->d_name[0] == '.') continue;
        snprintf(path, sizeof path, "/proc/self/fd/%s", ent->d_name);
        const ssize_t len = readlink(path, target, sizeof target - 1);
        if (len <= 0) continue;
        target[len] = 0;
        if (!strcmp(target, "/dev/kfd")) {
            found = true;
            break;
        }
    }
    closedir(d);
    return found;
}

__attribute__((constructor)) void runtime_defaults()
{
    const char *keep = getenv("FPNG_AMD_KEEP_HW_QUEUES");
    if (const char *cur = getenv("GPU_MAX_HW_QUEUES")) {
        g_hw_queues = std::max(1, atoi(cur));
        g_hw_queue_source = FPNG_AMD_HWQ_CALLER_SET;
    } else if (keep && keep[0] == '1') {
        g_hw_queue_source = FPNG_AMD_HWQ_HANDS_OFF;
    } else if (gpu_driver_open()) {
        g_hw_queue_source = FPNG_AMD_HWQ_DRIVER_OPEN; // the runtime has read its settings: its default of four stands
    } else {
        setenv("GPU_MAX_HW_QUEUES", "8", 0);
        g_hw_queues = 8;
        g_hw_queue_source = FPNG_AMD_HWQ_LIBRARY_SET;
    }
}
}

int fpng_amd::fail(int code, const char *what, hipError_t e)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    g_last_error = buf;
    return code;
}

namespace {

// per-device immutable tables
struct DeviceTables {
    TokenTable *one_pass[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // index by num_chans
    TokenTable *symbols[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // histogram pass: chunk[q] = length symbol - 256
    CrcDeviceTables *crc = nullptr;
    bool ready = false;
};
constexpr int kMaxDevices = 64;
DeviceTables g_dev[kMaxDevices];
TokenTable g_host_1pass[5];
uint32_t g_1pass_bits_per_byte[5]; // longest literal code (and no run token needs more per byte it covers)
std::mutex g_mu;
std::once_flag g_host_once;
bool g_host_ok = false;

// thread-safe lazy init: the fpng:: drop-in is re-entrant like the reference
bool host_tables()
{
    std::call_once(g_host_once, [] {
        if (!build_1pass_tables(&g_host_1pass[3], &g_host_1pass[4])) return;
        for (int c = 3; c <= 4; c++) {
            uint32_t m = 0;
            for (int i = 0; i < 257; i++) m = std::max(m, g_host_1pass[c].lit[i] >> 16);
            const uint32_t cap = (c == 3) ? kMaxChunkPixels3 : kMaxChunkPixels4;
            for (uint32_t q = 1; q <= cap; q++) m = std::max(m, ((g_host_1pass[c].chunk[q] >> 24) + q * c - 1) / (q * c)); // bits per covered byte
            g_1pass_bits_per_byte[c] = m;
        }
        g_host_ok = true;
    });
    return g_host_ok;
}

int ensure_device_tables(int dev)
{
    std::lock_guard<std::mutex> lock(g_mu);
    if (dev < 0 || dev >= kMaxDevices) return fail(FPNG_AMD_ERR_INVALID_ARG, "device index out of range");
    if (g_dev[dev].ready) return FPNG_AMD_OK;
    if (!host_tables()) return fail(FPNG_AMD_ERR_UNSUPPORTED, "format tables failed self-check");
    DeviceTables &d = g_dev[dev];
    for (int c = 3; c <= 4; c++) {
        HIP_TRY(hipMalloc(&d.one_pass[c], sizeof(TokenTable)));
        HIP_TRY(hipMemcpy(d.one_pass[c], &g_host_1pass[c], sizeof(TokenTable), hipMemcpyHostToDevice));
        TokenTable sym;
        std::memset(&sym, 0, sizeof sym);
        const uint32_t cap = (c == 3) ? kMaxChunkPixels3 : kMaxChunkPixels4;
        for (uint32_t q = 1; q <= cap; q++) {
            uint32_t s, e;
            deflate_length_symbol(q * c - 3, &s, &e);
            sym.chunk[q] = (s - 256) | (e << 8) | (((q * c - 3) & ((1u << e) - 1u)) << 16);
        }
        HIP_TRY(hipMalloc(&d.symbols[c], sizeof(TokenTable)));
        HIP_TRY(hipMemcpy(d.symbols[c], &sym, sizeof(TokenTable), hipMemcpyHostToDevice));
    }
    static CrcDeviceTables host_crc;
    build_crc_device_tables(&host_crc);
    HIP_TRY(hipMalloc(&d.crc, sizeof(CrcDeviceTables)));
    HIP_TRY(hipMemcpy(d.crc, &host_crc, sizeof(CrcDeviceTables), hipMemcpyHostToDevice));
    d.ready = true;
    return FPNG_AMD_OK;
}

} // namespace

// reference src/fpng.cpp:1670-1680 plus the 32-bit arithmetic limit of :1682-1705
int fpng_amd::check_dims(uint32_t w, uint32_t h, uint32_t c)
{
    if (w < 1 || h < 1 || (uint64_t)w * h > 0xFFFFFFFFull || w > (1u << 24) || h > (1u << 24))
        return fail(FPNG_AMD_ERR_INVALID_ARG, "invalid image dimensions");
    if (c != 3 && c != 4) return fail(FPNG_AMD_ERR_INVALID_ARG, "num_chans must be 3 or 4");
    if (((uint64_t)w * c + 1) * h > 0xFFFFFF00ull)
        return fail(FPNG_AMD_ERR_UNSUPPORTED, "more than 4 GiB of filtered bytes (undefined in the reference)");
    return FPNG_AMD_OK;
}

void fpng_amd::make_png_header(uint8_t *hdr, uint32_t w, uint32_t h, uint32_t c)
{
    // reference src/fpng.cpp:1767-1791.  Only the low 16 bits of each dimension are stored there
    // (":1773-1774"); reproduced, not fixed.  The IDAT length (bytes 50..53) is patched on device.
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x47, 0x0D, 0x0A, 0x1A, 0x0A};
    static const uint8_t fdec[17] = {0, 0, 0, 5, 'f', 'd', 'E', 'C', 82, 36, 147, 227, 0, 0xE5, 0xAB, 0x62, 0x99};
    std::memset(hdr, 0, 60);
    std::memcpy(hdr, sig, 8);
    hdr[11] = 13;
    std::memcpy(hdr + 12, "IHDR", 4);
    hdr[18] = (uint8_t)(w >> 8), hdr[19] = (uint8_t)w;
    hdr[22] = (uint8_t)(h >> 8), hdr[23] = (uint8_t)h;
    hdr[24] = 8;
    hdr[25] = (c == 3) ? 2 : 6;
    const uint32_t crc = host_crc32(hdr + 12, 17, 0);
    hdr[29] = (uint8_t)(crc >> 24), hdr[30] = (uint8_t)(crc >> 16), hdr[31] = (uint8_t)(crc >> 8), hdr[32] = (uint8_t)crc;
    std::memcpy(hdr + 33, fdec, 17);
    std::memcpy(hdr + 54, "IDAT", 4);
}

const TokenTable *fpng_amd::host_1pass_table(uint32_t c)
{
    return ((c == 3 || c == 4) && host_tables()) ? &g_host_1pass[c] : nullptr;
}

// ---- the list of large device blocks that are kept for the life of the process (encoder.h: DeviceBuf) ----
namespace {
struct CachedBlock {
    void *p;
    size_t bytes;
    int device;
};
std::mutex g_cache_mutex;
std::vector<CachedBlock> g_cache;
constexpr size_t kCacheMinBytes = 1u << 20;
} // namespace

void *fpng_amd::device_take(size_t bytes, size_t *got)
{
    if (bytes < kCacheMinBytes) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    size_t best = g_cache.size();
    for (size_t i = 0; i < g_cache.size(); i++) // the smallest block that fits and is not more than twice the request
        if (g_cache[i].device == dev && g_cache[i].bytes >= bytes && g_cache[i].bytes <= 2 * bytes &&
            (best == g_cache.size() || g_cache[i].bytes < g_cache[best].bytes))
            best = i;
    if (best == g_cache.size()) return nullptr;
    void *p = g_cache[best].p;
    *got = g_cache[best].bytes;
    g_cache.erase(g_cache.begin() + (long)best);
    return p;
}

void fpng_amd::device_give(void *p, size_t bytes)
{
    int dev = 0;
    if (bytes < kCacheMinBytes || hipGetDevice(&dev) != hipSuccess) {
        (void)hipFree(p);
        return;
    }
    (void)hipDeviceSynchronize(); // nothing in flight may still use it when its next owner writes to it
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    g_cache.push_back({p, bytes, dev});
}

extern "C" {

int fpng_amd_abi_version(void) { return FPNG_AMD_ABI_VERSION; }
const char *fpng_amd_last_error(void) { return g_last_error.c_str(); }

int fpng_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fpng_amd_device_available(void) { return fpng_amd_device_count() > 0 ? 1 : 0; }

int fpng_amd_release_cached_memory(void)
{
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &b : g_cache) {
        (void)hipSetDevice(b.device);
        (void)hipFree(b.p);
    }
    g_cache.clear();
    (void)hipSetDevice(cur);
    return FPNG_AMD_OK;
}

int fpng_amd_init(int device)
{
    if (!host_tables()) return fail(FPNG_AMD_ERR_UNSUPPORTED, "format tables failed self-check");
    if (fpng_amd_device_count() <= 0) return fail(FPNG_AMD_ERR_NO_DEVICE, "no HIP device");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    return ensure_device_tables(cur);
}

uint32_t fpng_amd_crc32(const void *data, size_t size, uint32_t prev) { return host_crc32(data, size, prev); }
uint32_t fpng_amd_adler32(const void *data, size_t size, uint32_t adler) { return host_adler32(data, size, adler); }
uint32_t fpng_amd_crc32_combine(uint32_t a, uint32_t b, uint64_t len_b) { return crc32_combine(a, b, len_b); }
uint32_t fpng_amd_adler32_combine(uint32_t a, uint32_t b, uint64_t len_b) { return adler32_combine(a, b, len_b); }

size_t fpng_amd_max_encoded_size(uint32_t w, uint32_t h, uint32_t c)
{
    const uint64_t n = ((uint64_t)w * c + 1) * h;
    return (size_t)(kPngHeaderBytes + 6 + n + 5 * ((n + kStoredBlockMax - 1) / kStoredBlockMax) + kPngTrailerBytes);
}

int fpng_amd_1pass_layout(uint32_t c, uint32_t *first_token_bit, uint32_t *eob_bits, uint32_t *prefix_bytes)
{
    if (c != 3 && c != 4) return fail(FPNG_AMD_ERR_INVALID_ARG, "num_chans must be 3 or 4");
    if (!host_tables()) return fail(FPNG_AMD_ERR_UNSUPPORTED, "format tables failed self-check");
    if (first_token_bit) *first_token_bit = g_host_1pass[c].first_token_bit;
    if (eob_bits) *eob_bits = g_host_1pass[c].lit[256] >> 16;
    if (prefix_bytes) *prefix_bytes = g_host_1pass[c].header_bits >> 3;
    return FPNG_AMD_OK;
}

static uint32_t default_lanes()
{
    const char *v = getenv("FPNG_AMD_LANES");
    const int n = v ? atoi(v) : (g_hw_queues >= 8 ? 4 : 2);
    return (uint32_t)std::min(std::max(n, 1), fpng_amd_encoder::kLanes);
}

int fpng_amd_runtime_info(fpng_amd_runtime *info)
{
    if (!info) return fail(FPNG_AMD_ERR_INVALID_ARG, "null runtime info");
    std::memset(info, 0, sizeof *info);
    info->hw_queues = (uint32_t)g_hw_queues;
    info->hw_queue_source = (uint32_t)g_hw_queue_source;
    info->lanes = default_lanes();
    return FPNG_AMD_OK;
}

uint32_t fpng_amd_encoder_lanes(const fpng_amd_encoder *e) { return e ? e->n_lanes : 0u; }

static int encoder_create(fpng_amd_encoder **out, int device, void *hip_stream, bool use_given_stream)
{
    if (!out) return fail(FPNG_AMD_ERR_INVALID_ARG, "null encoder pointer");
    *out = nullptr;
    if (fpng_amd_device_count() <= 0) return fail(FPNG_AMD_ERR_NO_DEVICE, "no HIP device");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    HIP_TRY(hipSetDevice(device));
    int rc = ensure_device_tables(device);
    if (rc) return rc;
    fpng_amd_encoder *e = new fpng_amd_encoder();
    e->device = device;
    if (use_given_stream) {
        e->stream = (hipStream_t)hip_stream; // nullptr = the legacy default stream
    } else {
        hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        if (err != hipSuccess) {
            delete e;
            return fail(FPNG_AMD_ERR_HIP, "hipStreamCreate", err);
        }
        e->own_stream = true;
    }
    e->n_lanes = default_lanes(); // fixed for the encoder's life: submit() takes turns over exactly the streams made here
    const uint32_t n_streams = std::max(e->n_lanes, 2u); // (the decoder uses the first two)
    for (uint32_t l = 0; l < n_streams; l++) {
        hipError_t err = hipStreamCreateWithFlags(&e->lane_stream[l], hipStreamNonBlocking);
        if (err != hipSuccess) {
            fpng_amd_encoder_destroy(e);
            return fail(FPNG_AMD_ERR_HIP, "hipStreamCreate (lane)", err);
        }
    }
    *out = e;
    return FPNG_AMD_OK;
}

int fpng_amd_encoder_create(fpng_amd_encoder **out, int device, void *hip_stream)
{
    return encoder_create(out, device, hip_stream, hip_stream != nullptr);
}

int fpng_amd_encoder_create_on_stream(fpng_amd_encoder **out, int device, void *hip_stream)
{
    return encoder_create(out, device, hip_stream, true);
}

int fpng_amd_encoder_set_stream(fpng_amd_encoder *e, void *hip_stream)
{
    if (!e) return fail(FPNG_AMD_ERR_INVALID_ARG, "null encoder");
    if (e->own_stream) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamDestroy(e->stream);
        e->own_stream = false;
    }
    e->stream = (hipStream_t)hip_stream;
    return FPNG_AMD_OK;
}

void fpng_amd_encoder_destroy(fpng_amd_encoder *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    destroy_host_workers(e);
    for (auto &ls : e->lane_stream)
        if (ls) (void)hipStreamSynchronize(ls);
    if (e->own_stream) (void)hipStreamSynchronize(e->stream);
    if (e->ev_ready)
        for (auto &ev : e->ev) (void)hipEventDestroy(ev);
    e->h_jobs.release();
    for (auto &sl : e->slots) {
        sl.jobs.release();
        sl.jobs2.release();
        sl.results.release();
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.in) (void)hipEventDestroy(sl.in);
        if (sl.walked) (void)hipEventDestroy(sl.walked);
    }
    for (auto &ls : e->lane_stream)
        if (ls) (void)hipStreamDestroy(ls);
    for (auto &ev : e->band_copied)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &s : e->sc) s.release();
    e->h_states.release();
    e->h_partials.release();
    e->d_stream_partials.release();
    e->d_decode.release();
    e->d_dec_gran.release();
    e->d_lut_cache.release();
    for (hipEvent_t ev : e->dec_prof_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->dec_ev2)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->dec_ev3)
        if (ev) (void)hipEventDestroy(ev);
    e->h_dec_fetch.release();
    if (e->dec_up) (void)hipStreamDestroy(e->dec_up);
    for (hipEvent_t ev : e->dec_ev)
        if (ev) (void)hipEventDestroy(ev);
    e->d_xchg.release();
    e->h_xchg.release();
    e->d_stage_in.release();
    e->d_stage_out.release();
    for (auto &b : e->host.d_in) b.release();
    for (auto &b : e->host.d_out) b.release();
    if (e->host.up) (void)hipStreamDestroy(e->host.up);
    if (e->host.down) (void)hipStreamDestroy(e->host.down);
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

void *fpng_amd_encoder_stream(fpng_amd_encoder *e) { return e ? (void *)e->stream : nullptr; }

int fpng_amd_encoder_set_profiling(fpng_amd_encoder *e, int enabled)
{
    if (!e) return fail(FPNG_AMD_ERR_INVALID_ARG, "null encoder");
    if (enabled && !e->ev_ready) {
        HIP_TRY(hipSetDevice(e->device));
        for (auto &ev : e->ev) HIP_TRY(hipEventCreate(&ev));
        e->ev_ready = true;
    }
    e->profiling = enabled != 0;
    return FPNG_AMD_OK;
}

int fpng_amd_encoder_last_phase_ms(fpng_amd_encoder *e, float ms[FPNG_AMD_NUM_PHASES])
{
    if (!e || !ms) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < FPNG_AMD_NUM_PHASES; i++) ms[i] = 0.f;
    if (!e->profiling || !e->phases_recorded) return FPNG_AMD_OK;
    HIP_TRY(hipEventSynchronize(e->ev[e->phases_recorded]));
    for (uint32_t i = 0; i < e->phases_recorded; i++) HIP_TRY(hipEventElapsedTime(&ms[i], e->ev[i], e->ev[i + 1]));
    return FPNG_AMD_OK;
}

} // extern "C"

namespace {

constexpr uint64_t kLocalFrontPad = 32; // dwords (one 128-byte line)

struct Submission {
    uint32_t n = 0, max_rows = 0, max_crc_blocks = 0;
    uint64_t total_rows = 0;
    uint64_t local_dwords = kLocalFrontPad; // scratch for the rows' local streams (assemble_kernel may read up to four dwords in front of a stream)
    uint32_t chan_mask = 0;    // bit 0: 3-channel jobs present, bit 1: 4-channel jobs
    uint64_t px4 = 0, px4_wide = 0; // pixels of the 4-channel jobs, and of those with rows of kWideRowPixels and more
};

int mark(fpng_amd_encoder *e, hipStream_t s, uint32_t idx)
{
    if (e->profiling) {
        HIP_TRY(hipEventRecord(e->ev[idx], s));
        e->phases_recorded = idx;
    }
    return FPNG_AMD_OK;
}

} // namespace

// Host-side wait for every submission in flight on the lanes.
int fpng_amd::drain(fpng_amd_encoder *e)
{
    for (auto &sl : e->slots)
        if (sl.in_flight) {
            HIP_TRY(hipEventSynchronize(sl.done));
            sl.in_flight = false;
        }
    return FPNG_AMD_OK;
}

namespace {

// Fills slot.jobs[0..n) for whole-image jobs and sizes the scratch buffers.  Only `slot` (which is free) and
// the lane's device scratch are touched: the records of submissions in flight stay where they are.
int prepare_jobs(fpng_amd_encoder *e, fpng_amd_encoder::Slot &slot, fpng_amd_encoder::Scratch &sc, const fpng_amd_image *images,
                 uint32_t n, uint32_t flags, Submission &sub)
{
    if (!e || !images || !n) return fail(FPNG_AMD_ERR_INVALID_ARG, "null/empty batch");
    if (n > 65535) return fail(FPNG_AMD_ERR_INVALID_ARG, "batch larger than 65535 images");
    int rc;
    if ((rc = slot.jobs.ensure(n)) || (rc = slot.results.ensure(n))) return rc;
    const DeviceTables &dt = g_dev[e->device];
    const bool force_stored = (flags & FPNG_AMD_FORCE_UNCOMPRESSED) != 0;
    const bool two_pass = (flags & FPNG_AMD_ENCODE_SLOWER) && !force_stored;
    sub = Submission();
    sub.n = n;
    for (uint32_t i = 0; i < n; i++) {
        const fpng_amd_image &im = images[i];
        if ((rc = check_dims(im.w, im.h, im.num_chans))) return rc;
        if (!im.d_pixels || !im.d_out) return fail(FPNG_AMD_ERR_INVALID_ARG, "null device pointer");
        if (((uintptr_t)im.d_out & 15) || (im.num_chans == 4 && ((uintptr_t)im.d_pixels & 3)))
            return fail(FPNG_AMD_ERR_INVALID_ARG, "d_out must be 16-byte aligned, RGBA d_pixels 4-byte aligned");
        if (im.out_cap < fpng_amd_max_encoded_size(im.w, im.h, im.num_chans))
            return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "out_cap < fpng_amd_max_encoded_size()");
        Job &j = slot.jobs.p[i];
        std::memset(&j, 0, sizeof j);
        j.rows = (const uint8_t *)im.d_pixels;
        j.row_above = nullptr;
        j.out = im.d_out;
        j.out_cap = im.out_cap;
        j.w = im.w;
        j.c = im.num_chans;
        j.bpl = im.w * im.num_chans;
        j.nrows = j.h_total = im.h;
        j.y0 = 0;
        j.flags = flags & (FPNG_AMD_ENCODE_SLOWER | FPNG_AMD_FORCE_UNCOMPRESSED); // (the reference looks at these two bits only; the bits above are the kernels' own)
        j.row_base = (uint32_t)sub.total_rows;
        j.one_pass = two_pass ? 0 : 1;
        j.whole_png = j.is_first = j.is_last = 1;
        j.bit_bias = (int64_t)kPngHeaderBytes * 8;
        j.table = two_pass ? nullptr /* patched below */ : dt.one_pass[im.num_chans];
        j.crc_blocks = (uint32_t)((fpng_amd_max_encoded_size(im.w, im.h, im.num_chans) + kCrcRangeBytes - 1) / kCrcRangeBytes) + 1;
        make_png_header(j.png_header, im.w, im.h, im.num_chans);
        // a row's local stream: at most L bits per filtered byte, L = the longest literal code of the table in use
        // (12 for the per-image tables of 2-pass, reference fpng.cpp:1111; run tokens need less per byte they
        // cover), + end-of-block; rows start 16-byte aligned and the 16-byte flush / assemble_kernel may touch up
        // to four dwords past the stream
        const uint64_t bits_per_byte = two_pass ? 12u : g_1pass_bits_per_byte[im.num_chans];
        j.local_stride = (uint32_t)(((((uint64_t)j.bpl + 1) * bits_per_byte + 64 + 31) / 32 + 4 + 31) & ~31ull); // whole 128-byte lines
        j.local_base = sub.local_dwords;
        sub.chan_mask |= (im.num_chans == 3) ? 1u : 2u;
        if (im.num_chans == 4) sub.px4 += (uint64_t)im.w * im.h, sub.px4_wide += im.w >= kWideRowPixels ? (uint64_t)im.w * im.h : 0u;
        const uint64_t units = im.h; // records of the job: one per row
        sub.local_dwords += (uint64_t)j.local_stride * units;
        if (sub.total_rows + units > 0xFFFFFFFFull) return fail(FPNG_AMD_ERR_UNSUPPORTED, "too many rows in one batch");
        sub.total_rows += units;
        sub.max_rows = std::max(sub.max_rows, im.h);
        sub.max_crc_blocks = std::max(sub.max_crc_blocks, j.crc_blocks);
    }
    if ((rc = sc.d_jobs.ensure(n))) return rc;
    if ((rc = sc.d_rows.ensure(sub.total_rows))) return rc;
    if ((rc = sc.d_row_off.ensure(sub.total_rows))) return rc;
    if ((rc = sc.d_states.ensure(n))) return rc;
    if ((rc = sc.d_results.ensure(n))) return rc;
    if ((rc = sc.d_partials.ensure(3 * (size_t)n * sub.max_crc_blocks))) return rc; // CRC partials + two Adler words per range (stored images)
    if (two_pass) {
        if ((rc = sc.d_hist.ensure((size_t)n * 288))) return rc;
        if ((rc = sc.d_dyn.ensure(n))) return rc;
    }
    return FPNG_AMD_OK;
}

int submit(fpng_amd_encoder *e, const fpng_amd_image *images, uint32_t n, uint32_t flags, uint64_t *ticket_out)
{
    if (!e) return fail(FPNG_AMD_ERR_INVALID_ARG, "null encoder");
    HIP_TRY(hipSetDevice(e->device));
    // next slot of the ring; wait only if the submission that used it is still running.  Nothing of the encoder's
    // state changes before the batch has been validated.
    fpng_amd_encoder::Slot &slot = e->slots[(e->submitted + 1) % fpng_amd_encoder::kSlots];
    if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    if (!slot.in) HIP_TRY(hipEventCreateWithFlags(&slot.in, hipEventDisableTiming));
    if (!slot.walked) HIP_TRY(hipEventCreateWithFlags(&slot.walked, hipEventDisableTiming));
    if (slot.in_flight) {
        HIP_TRY(hipEventSynchronize(slot.done));
        slot.in_flight = false;
    }
    const uint32_t n_lanes = e->n_lanes;
    // lane = internal stream + scratch set; whole chains alternate between FPNG_AMD_LANES (default: four over eight hardware queues, else two) of them, so that the
    // latency-bound tail of one submission (scan, stored fallback, finalize) and its memory-bound assemble run next to the
    // row walk of the next one.  Per-kernel profiling stays on lane 0, which serialises it.
    // (Measured on one box and dropped: the three stages of a submission on three streams linked by events -- prep, walk,
    // place, the latter with high priority -- so that walks follow each other without a gap: 0.599 vs 0.555 ms per 8 x 8K
    // step, +2 % for 2-pass only.)
    const int lane = e->profiling ? 0 : (int)(e->submitted % n_lanes);
    hipStream_t s = e->lane_stream[lane];
    fpng_amd_encoder::Scratch &sc = e->sc[lane];
    Submission sub;
    int rc = prepare_jobs(e, slot, sc, images, n, flags, sub);
    if (rc) return rc;
    const DeviceTables &dt = g_dev[e->device];
    const bool force_stored = (flags & FPNG_AMD_FORCE_UNCOMPRESSED) != 0;
    const bool two_pass = (flags & FPNG_AMD_ENCODE_SLOWER) && !force_stored;
    // Every row is encoded into its own scratch stream ("local stream"), a second kernel shifts the streams into place and
    // takes the CRC.  The scratch is sized for the worst case of the table in use (FPNG_AMD_LOCAL_LIMIT_MB caps it).
    static const uint64_t local_limit = [] {
        const char *v = getenv("FPNG_AMD_LOCAL_LIMIT_MB");
        return (v ? (uint64_t)atoll(v) : 98304ull) << 20;
    }();
    if ((sub.local_dwords + 16) * 4 > local_limit)
        return fail(FPNG_AMD_ERR_OUT_OF_MEMORY, "scratch for the rows' local streams exceeds FPNG_AMD_LOCAL_LIMIT_MB: submit fewer images per call");
    if ((rc = sc.d_local.ensure(sub.local_dwords + 16))) return rc;
    for (uint32_t i = 0; i < n; i++) slot.jobs.p[i].flags |= 0x200u;
    if (two_pass) {
        // pass 1 works on the symbol table; the per-job dynamic table is built on device.  The second
        // job array (same jobs, pointing at their dynamic tables) is prepared now so nothing waits later.
        if ((rc = slot.jobs2.ensure(n))) return rc;
        for (uint32_t i = 0; i < n; i++) {
            slot.jobs2.p[i] = slot.jobs.p[i];
            slot.jobs2.p[i].table = sc.d_dyn.p + i;
            slot.jobs.p[i].table = dt.symbols[slot.jobs.p[i].c];
        }
    }
    // ---- the batch is valid.  The ticket is committed at the very end, after the last HIP call that can fail: a failed
    //      submit hands out no ticket and the next one reuses this slot and lane (stream order keeps whatever was launched
    //      apart from it) ----
    slot.ticket = 0;
    e->phases_recorded = 0;
    e->last_two_pass = two_pass;
    // everything the caller enqueued on the encoder's stream so far (e.g. the producer of the pixels)
    // (when that stream has nothing pending there is nothing to order against: no marker + barrier packet in front of the chain;
    // the same for the scratch set's previous user when it is done already)
    if (hipStreamQuery(e->stream) != hipSuccess) {
        (void)hipGetLastError(); // ("not ready" is not an error)
        HIP_TRY(hipEventRecord(slot.in, e->stream));
        HIP_TRY(hipStreamWaitEvent(s, slot.in, 0));
    }
    if (sc.last_done && hipEventQuery(sc.last_done) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipStreamWaitEvent(s, sc.last_done, 0)); // the scratch set's previous user
    }
    // (Measured and dropped, profiles/r03_latency.txt: letting one- and two-frame submissions read their job records straight from
    // the slot's pinned host memory instead of uploading them -- the upload is a blit kernel + a dispatch gap, ~7 us -- costs
    // 7 us MORE per chain: every kernel's first touch of the record goes over PCIe.)
    // One image: its record travels in the arguments of the chain's first kernel, which leaves it in d_jobs for the others
    // (encode_rows_first_kernel): no blit kernel + dispatch gap in front of the chain.
    const bool job_in_args = n == 1 && !force_stored;
    const Job *d_jobs = sc.d_jobs.p;
    if (!job_in_args) HIP_TRY(hipMemcpyAsync(sc.d_jobs.p, slot.jobs.p, n * sizeof(Job), hipMemcpyHostToDevice, s));
    if ((rc = mark(e, s, 0))) return rc;
    uint32_t ph = 0; // index of the last phase mark
    if (two_pass) {
        // the counters must start at zero: the table builder leaves the ones it consumed zeroed, so only a fresh (or otherwise
        // used, or too short) buffer is cleared here -- a fill kernel + dispatch gap less in front of a 2-pass chain
        const size_t n_counters = (size_t)n * 288;
#ifdef FPNG_BUILD_TIMING
        const uint32_t rezero = 0; // (the timing build leaves its cycle counts in the histogram buffer)
#else
        const uint32_t rezero = 1;
#endif
        if (sc.d_hist.fresh) sc.hist_zero = 0, sc.d_hist.fresh = false;
        if (sc.hist_zero < n_counters) HIP_TRY(hipMemsetAsync(sc.d_hist.p, 0, n_counters * sizeof(uint32_t), s));
        sc.hist_zero = 0;
        if (job_in_args)
            launch_hist_first(s, slot.jobs.p[0], sc.d_jobs.p, sc.d_hist.p);
        else
            launch_hist(s, sc.d_jobs.p, n, sub.max_rows, sc.d_hist.p);
        if ((rc = mark(e, s, ++ph))) return rc;
        launch_build_dynamic(s, sc.d_jobs.p, n, sc.d_hist.p, sc.d_dyn.p, rezero);
        if (rezero) sc.hist_zero = n_counters;
        if (!job_in_args) HIP_TRY(hipMemcpyAsync(sc.d_jobs.p, slot.jobs2.p, n * sizeof(Job), hipMemcpyHostToDevice, s));
        if ((rc = mark(e, s, ++ph))) return rc;
    }
    // 2-pass only: the row walk of this submission waits for the walk of the previous one (other lane), so that
    // its own histogram pass and table build run under that walk instead of next to the other lane's (+9 %, measured;
    // 1-pass loses 4 % with the same rule).  FPNG_AMD_STAGGER=0/1 forces it off/on for A/B runs.
    static const int stagger_env = [] {
        const char *v = getenv("FPNG_AMD_STAGGER");
        return v ? (v[0] == '1' ? 1 : 0) : -1;
    }();
    // (With four lanes over eight hardware queues the rule turns into a loss -- 8 x 8K 2-pass 0.622 -> 0.591 ms per step without it,
    // profiles/r05_hw_queues.txt -- so it stays with the two-lane case it was measured for.)
    const bool stagger = stagger_env < 0 ? (two_pass && n_lanes <= 2) : stagger_env == 1;
    if (stagger && e->prev_walked && !e->profiling) HIP_TRY(hipStreamWaitEvent(s, e->prev_walked, 0));
    // four launches: rows, row scan (sizes, offsets, stored-or-compressed decision, stream head), assemble (rows into place or,
    // for an image that fell back, the stored blocks; + CRC partials), finalize (CRC fold, Adler, trailer, result record).  Folding the scan into the last row block and
    // the finalize step into the last assemble block (three launches) was measured on the same box: 11 % less throughput
    // (both kernels get slower by more than the two small launches cost) and 5-19 % MORE single-frame latency.
    if (job_in_args)
        launch_encode_rows_first(s, two_pass ? slot.jobs2.p[0] : slot.jobs.p[0], sc.d_jobs.p, sc.d_rows.p, sc.d_states.p, sc.d_local.p);
    else if (!force_stored)
        launch_encode_rows(s, d_jobs, n, sub.max_rows, sub.chan_mask, sc.d_rows.p, sc.d_states.p, sc.d_local.p, 2 * sub.px4_wide >= sub.px4);
    if ((rc = mark(e, s, ++ph))) return rc;
    if (stagger) { // (only staggered walks wait for it)
        // ... and only a submission that follows while this one runs: with every other lane idle (one frame at a time) the marker
        // packet between the walk and the scan would only lengthen the chain
        bool others_busy = false;
        for (uint32_t l = 0; l < n_lanes && !others_busy; l++)
            if ((int)l != lane && e->sc[l].last_done && hipEventQuery(e->sc[l].last_done) != hipSuccess) others_busy = true;
        (void)hipGetLastError();
        e->prev_walked = nullptr;
        if (others_busy) {
            HIP_TRY(hipEventRecord(slot.walked, s));
            e->prev_walked = slot.walked;
        }
    }
    launch_scan(s, d_jobs, n, sc.d_rows.p, sc.d_row_off.p, sc.d_states.p);
    if ((rc = mark(e, s, ++ph))) return rc;
    uint32_t *adler_parts = sc.d_partials.p + (size_t)n * sub.max_crc_blocks;
    launch_assemble(s, d_jobs, n, sub.max_crc_blocks, sc.d_states.p, sc.d_row_off.p, sc.d_local.p, dt.crc, sc.d_partials.p, adler_parts);
    if ((rc = mark(e, s, ++ph))) return rc;
    launch_finalize(s, d_jobs, n, sub.max_crc_blocks, sc.d_rows.p, sc.d_states.p, dt.crc, sc.d_partials.p, adler_parts, slot.results.p);
    if ((rc = mark(e, s, ++ph))) return rc;
    // the result records go straight into the slot's pinned host memory (device-visible): no copy kernel at the
    // end of the chain; they are read by the host after the `done` event
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(slot.done, s));
    sc.last_done = slot.done;
    e->submitted++;
    slot.ticket = e->submitted;
    slot.n = n;
    slot.in_flight = true;
    if (ticket_out) *ticket_out = slot.ticket;
    return FPNG_AMD_OK;
}

fpng_amd_encoder::Slot *slot_of(fpng_amd_encoder *e, uint64_t ticket)
{
    if (!ticket || ticket > e->submitted || ticket + fpng_amd_encoder::kSlots <= e->submitted) return nullptr;
    fpng_amd_encoder::Slot &sl = e->slots[ticket % fpng_amd_encoder::kSlots];
    return sl.ticket == ticket ? &sl : nullptr;
}

} // namespace

extern "C" {

int fpng_amd_encode_submit(fpng_amd_encoder *e, const fpng_amd_image *images, uint32_t n, uint32_t flags, uint64_t *ticket)
{
    return submit(e, images, n, flags, ticket);
}

int fpng_amd_encode_batch_async(fpng_amd_encoder *e, const fpng_amd_image *images, uint32_t n, uint32_t flags)
{
    return submit(e, images, n, flags, nullptr);
}

int fpng_amd_encode_query(fpng_amd_encoder *e, uint64_t ticket)
{
    if (!e) return fail(FPNG_AMD_ERR_INVALID_ARG, "null encoder");
    fpng_amd_encoder::Slot *sl = slot_of(e, ticket);
    if (!sl) return fail(FPNG_AMD_ERR_INVALID_ARG, "unknown or expired ticket");
    if (!sl->in_flight) return 1;
    const hipError_t st = hipEventQuery(sl->done);
    if (st == hipSuccess) return 1;
    if (st == hipErrorNotReady) return 0;
    return fail(FPNG_AMD_ERR_HIP, "hipEventQuery", st);
}

int fpng_amd_encode_wait(fpng_amd_encoder *e, uint64_t ticket, fpng_amd_result *results, uint32_t n)
{
    if (!e) return fail(FPNG_AMD_ERR_INVALID_ARG, "null encoder");
    HIP_TRY(hipSetDevice(e->device));
    fpng_amd_encoder::Slot *sl = slot_of(e, ticket);
    if (!sl) return fail(FPNG_AMD_ERR_INVALID_ARG, "unknown or expired ticket (records are kept for the last 8 submissions)");
    if (sl->in_flight) {
        HIP_TRY(hipEventSynchronize(sl->done));
        sl->in_flight = false;
    }
    if (results) {
        if (n > sl->n) return fail(FPNG_AMD_ERR_INVALID_ARG, "more results requested than images submitted");
        for (uint32_t i = 0; i < n; i++) {
            results[i].png_size = sl->results.p[i].png_size;
            results[i].mode = sl->results.p[i].mode;
            results[i].status = sl->results.p[i].status;
        }
    }
    return FPNG_AMD_OK;
}

const char *fpng_amd_encoder_phase_names(fpng_amd_encoder *e)
{
    return (e && e->last_two_pass) ? "hist,build_dynamic,encode_rows,scan,assemble,finalize" : "encode_rows,scan,assemble,finalize";
}

int fpng_amd_encoder_join(fpng_amd_encoder *e)
{
    if (!e) return fail(FPNG_AMD_ERR_INVALID_ARG, "null encoder");
    HIP_TRY(hipSetDevice(e->device));
    for (auto &sl : e->slots)
        if (sl.in_flight) HIP_TRY(hipStreamWaitEvent(e->stream, sl.done, 0));
    return FPNG_AMD_OK;
}

int fpng_amd_encode_finish(fpng_amd_encoder *e, fpng_amd_result *results, uint32_t n)
{
    if (!e) return fail(FPNG_AMD_ERR_INVALID_ARG, "null encoder");
    HIP_TRY(hipSetDevice(e->device));
    int rc0 = drain(e); // every outstanding submission is done after this
    if (rc0) return rc0;
    if (results) {
        if (!e->submitted) return fail(FPNG_AMD_ERR_INVALID_ARG, "nothing was submitted");
        return fpng_amd_encode_wait(e, e->submitted, results, n);
    }
    return FPNG_AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// Host-buffer front door for MANY frames: a ring of device staging buffers and three actors -- an uploader
// thread (H2D of frame k+1), the calling thread (encode submissions, one per frame, each ordered behind its upload) and a
// downloader thread (D2H of frame k, then the optional file write on a pool of writer threads) -- so that the two PCIe
// directions and the encoder overlap instead of taking turns as in fpng_amd_encode_host().
// ------------------------------------------------------------------------------------------------
static int host_batch_ring(fpng_amd_encoder *e, const fpng_amd_host_image *imgs, uint32_t n, uint32_t flags, int n_writer_threads);

int fpng_amd_encode_host_batch(fpng_amd_encoder *e, const fpng_amd_host_image *imgs, uint32_t n, uint32_t flags, int n_writer_threads)
{
    if (!e || !imgs || !n) return fail(FPNG_AMD_ERR_INVALID_ARG, "null/empty batch");
    int rc;
    for (uint32_t i = 0; i < n; i++) {
        if ((rc = check_dims(imgs[i].w, imgs[i].h, imgs[i].num_chans))) return rc;
        if (!imgs[i].pixels || (!imgs[i].out && !imgs[i].path)) return fail(FPNG_AMD_ERR_INVALID_ARG, "image without pixels or destination");
    }
    return host_batch_ring(e, imgs, n, flags, n_writer_threads);
}

static int host_batch_ring(fpng_amd_encoder *e, const fpng_amd_host_image *imgs, uint32_t n, uint32_t flags, int n_writer_threads)
{
    int rc;
    HIP_TRY(hipSetDevice(e->device));
    if ((rc = drain(e))) return rc;
    using HostRing = fpng_amd_encoder::HostRing;
    HostRing &ring = e->host; // (staging buffers are kept between calls)
    if ((rc = ensure_copy_streams(e))) return rc;
    size_t max_in = 0, max_out = 0;
    for (uint32_t i = 0; i < n; i++) {
        max_in = std::max(max_in, (size_t)imgs[i].w * imgs[i].h * imgs[i].num_chans);
        max_out = std::max(max_out, fpng_amd_max_encoded_size(imgs[i].w, imgs[i].h, imgs[i].num_chans));
    }
    // ring depth: three big frames are plenty (each stage takes milliseconds), small frames need more slots in flight to hide the
    // fixed costs of their three stages (a 512 x 512 frame: ~130 us from upload to download against 15 us on the link);
    // at most kSlots - 1 submissions are in flight anyway
    const int depth = (int)std::min<size_t>(std::min<size_t>(HostRing::kDepth, n), std::max<size_t>(3, (size_t)(48u << 20) / std::max<size_t>(max_in, 1)));
    for (int k = 0; k < depth; k++)
        if ((rc = ring.d_in[k].ensure(max_in + 16)) || (rc = ring.d_out[k].ensure(max_out + 64))) return rc;

    // slot k is handed round: uploader (state 0 -> 1), caller submits (1 -> 2), downloader frees it (2 -> 0)
    std::mutex mu;
    std::condition_variable cv;
    int state[HostRing::kDepth] = {};
    uint64_t tickets[HostRing::kDepth] = {};
    std::mutex emu; // the encoder object is not thread-safe: the submitting thread and the downloader take turns
    std::atomic<int> failed{0};
    const int device = e->device;
    std::vector<size_t> sizes(n, 0);
    std::vector<std::vector<uint8_t>> file_bufs(n); // frames that go to files without a caller buffer (freed once written)
    // writer pool: finished files are handed to n_writer_threads threads (reference fpng.cpp:1806-1828 writes inline)
    struct WriteJob { const char *path; const uint8_t *data; size_t size; uint32_t idx; };
    std::vector<WriteJob> wq;
    size_t wq_head = 0, wq_done = 0; // taken / written; at most 2 x writers + 2 frames wait in between (back-pressure on the downloader)
    bool wq_closed = false;
    std::mutex wmu;
    std::condition_variable wcv;
    std::vector<std::thread> writers;
    const int nw = std::max(0, std::min(n_writer_threads, 16));
    for (int t = 0; t < nw; t++)
        writers.emplace_back([&] {
            for (;;) {
                WriteJob job;
                {
                    std::unique_lock<std::mutex> lk(wmu);
                    wcv.wait(lk, [&] { return wq_head < wq.size() || wq_closed; });
                    if (wq_head >= wq.size()) return;
                    job = wq[wq_head++];
                }
                FILE *f = fopen(job.path, "wb");
                if (!f || fwrite(job.data, 1, job.size, f) != job.size) failed = FPNG_AMD_ERR_IO;
                if (f && fclose(f) == EOF) failed = FPNG_AMD_ERR_IO;
                std::vector<uint8_t>().swap(file_bufs[job.idx]); // (a frame without a caller buffer: its bytes are on disk now)
                {
                    std::lock_guard<std::mutex> lk(wmu);
                    wq_done++;
                }
                wcv.notify_all(); // the downloader may be waiting for room in the queue; a failure is seen at once
            }
        });

    std::thread uploader([&] {
        (void)hipSetDevice(device);
        for (uint32_t i = 0; i < n && !failed; i++) {
            const int k = (int)(i % (uint32_t)depth);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return state[k] == 0 || failed; });
            }
            if (failed) break;
            const size_t bytes = (size_t)imgs[i].w * imgs[i].h * imgs[i].num_chans;
            if (hipMemcpyAsync(ring.d_in[k].p, imgs[i].pixels, bytes, hipMemcpyHostToDevice, ring.up) != hipSuccess ||
                hipStreamSynchronize(ring.up) != hipSuccess)
                failed = FPNG_AMD_ERR_HIP;
            {
                std::lock_guard<std::mutex> lk(mu);
                state[k] = 1;
            }
            cv.notify_all();
        }
    });
    std::thread downloader([&] {
        (void)hipSetDevice(device);
        for (uint32_t i = 0; i < n && !failed; i++) {
            const int k = (int)(i % (uint32_t)depth);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return state[k] == 2 || failed; });
            }
            if (failed) break;
            // wait for THIS frame's submission on its `done` event, outside the encoder lock (the next submit must not wait)
            fpng_amd_result res = {0, 0, 1};
            hipEvent_t done = nullptr;
            fpng_amd_encoder::Slot *sl = nullptr;
            {
                std::lock_guard<std::mutex> lk(emu);
                sl = slot_of(e, tickets[k]);
                if (sl) done = sl->done;
            }
            if (sl && hipEventSynchronize(done) == hipSuccess) {
                std::lock_guard<std::mutex> lk(emu);
                res.png_size = sl->results.p[0].png_size;
                res.mode = sl->results.p[0].mode;
                res.status = sl->results.p[0].status;
                sl->in_flight = false;
            }
            if (res.status) {
                failed = FPNG_AMD_ERR_HIP;
            } else {
                uint8_t *dst = imgs[i].out;
                if (!dst) {
                    file_bufs[i].resize(res.png_size);
                    dst = file_bufs[i].data();
                } else if (imgs[i].out_cap < res.png_size) {
                    failed = FPNG_AMD_ERR_BUFFER_TOO_SMALL;
                }
                if (!failed && (hipMemcpyAsync(dst, ring.d_out[k].p, res.png_size, hipMemcpyDeviceToHost, ring.down) != hipSuccess ||
                                hipStreamSynchronize(ring.down) != hipSuccess))
                    failed = FPNG_AMD_ERR_HIP;
                sizes[i] = (size_t)res.png_size;
                if (!failed && imgs[i].path) {
                    if (nw) {
                        {
                            // a disk slower than the GPU must not let finished frames pile up in host memory
                            std::unique_lock<std::mutex> lk(wmu);
                            wcv.wait(lk, [&] { return wq.size() - wq_done < (size_t)(2 * nw + 2) || failed; });
                            wq.push_back({imgs[i].path, dst, (size_t)res.png_size, i});
                        }
                        wcv.notify_all();
                    } else {
                        FILE *f = fopen(imgs[i].path, "wb");
                        if (!f || fwrite(dst, 1, res.png_size, f) != res.png_size) failed = FPNG_AMD_ERR_IO;
                        if (f && fclose(f) == EOF) failed = FPNG_AMD_ERR_IO;
                        std::vector<uint8_t>().swap(file_bufs[i]);
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                state[k] = 0;
            }
            cv.notify_all();
        }
    });
    // the calling thread: one encode submission per frame, ordered behind that frame's upload
    for (uint32_t i = 0; i < n && !failed; i++) {
        const int k = (int)(i % (uint32_t)depth);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return state[k] == 1 || failed; });
        }
        if (failed) break;
        fpng_amd_image im;
        im.d_pixels = ring.d_in[k].p;
        im.w = imgs[i].w, im.h = imgs[i].h, im.num_chans = imgs[i].num_chans;
        im.d_out = ring.d_out[k].p;
        im.out_cap = ring.d_out[k].cap;
        // (the upload was waited for on the host: nothing more to order the submission against)
        uint64_t t = 0;
        {
            std::lock_guard<std::mutex> lk(emu);
            if ((rc = submit(e, &im, 1, flags, &t))) failed = rc;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            tickets[k] = t;
            state[k] = 2;
        }
        cv.notify_all();
    }
    if (failed) cv.notify_all();
    uploader.join();
    downloader.join();
    {
        std::lock_guard<std::mutex> lk(wmu);
        wq_closed = true;
    }
    wcv.notify_all();
    for (auto &t : writers) t.join();
    for (uint32_t i = 0; i < n; i++)
        if (imgs[i].out_size) *imgs[i].out_size = sizes[i];
    if (failed) return fail(failed, "host batch failed (copy, encode or file write)");
    return FPNG_AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// row bands: one image sharded by rows over several GPUs.  Per band: [histogram ->] encode (rows into
// the encoder's local streams, counts to the host: the one exchange step) -> place (the streams shifted to the band's
// bit position inside a window that shares the file's 16-byte geometry).  Same kernels as whole images.
// ------------------------------------------------------------------------------------------------
// number of 64 KiB CRC ranges of a file whose zlib stream has zlib_size bytes (ranges end at the 16-byte-aligned end of the
// data; the same arithmetic as finalize_kernel)
static uint32_t crc_ranges_of(uint64_t zlib_size)
{
    const int64_t data_end = (int64_t)(kPngHeaderBytes + zlib_size - 4);
    const int64_t end_aligned = (data_end + 15) & ~15ll;
    return (uint32_t)((end_aligned - 48 + (1ll << 16) - 1) >> 16);
}

static int band_check(fpng_amd_encoder *e, const fpng_amd_band *b)
{
    if (!e || !b || !b->d_rows) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    if (b->y1 <= b->y0 || b->y1 > b->h_total) return fail(FPNG_AMD_ERR_INVALID_ARG, "empty or out-of-range band");
    int rc = check_dims(b->w, b->h_total, b->num_chans);
    if (rc) return rc;
    if (b->y0 > 0 && !b->d_row_above) return fail(FPNG_AMD_ERR_INVALID_ARG, "band needs the row above its first row");
    if (b->num_chans == 4 && (((uintptr_t)b->d_rows & 3) || ((uintptr_t)b->d_row_above & 3)))
        return fail(FPNG_AMD_ERR_INVALID_ARG, "RGBA rows must be 4-byte aligned");
    return FPNG_AMD_OK;
}

// The asynchronous band calls upload their job record from pinned memory: before record k is rewritten, the previous
// upload from it must have happened.
static int band_record_free(fpng_amd_encoder *e, int k)
{
    if (!e->band_copied[k])
        HIP_TRY(hipEventCreateWithFlags(&e->band_copied[k], hipEventDisableTiming));
    else
        HIP_TRY(hipEventSynchronize(e->band_copied[k]));
    return FPNG_AMD_OK;
}

static void band_job(fpng_amd_encoder *e, const fpng_amd_band *b, bool two_pass, Job &j)
{
    std::memset(&j, 0, sizeof j);
    j.rows = (const uint8_t *)b->d_rows;
    j.row_above = (const uint8_t *)b->d_row_above;
    j.w = b->w, j.c = b->num_chans, j.bpl = b->w * b->num_chans;
    j.nrows = b->y1 - b->y0;
    j.y0 = b->y0;
    j.h_total = b->h_total;
    j.one_pass = two_pass ? 0 : 1;
    j.is_first = b->y0 == 0;
    j.is_last = b->y1 == b->h_total;
    j.bit_bias = (int64_t)kPngHeaderBytes * 8;
    const uint64_t bits_per_byte = two_pass ? 12u : g_1pass_bits_per_byte[b->num_chans];
    j.local_stride = (uint32_t)(((((uint64_t)j.bpl + 1) * bits_per_byte + 64 + 31) / 32 + 4 + 31) & ~31ull); // whole 128-byte lines
    j.local_base = kLocalFrontPad;
    j.table = two_pass ? e->sc[0].d_dyn.p : g_dev[e->device].one_pass[b->num_chans];
}

int fpng_amd_band_hist(fpng_amd_encoder *e, const fpng_amd_band *b, uint32_t *d_hist288)
{
    int rc = band_check(e, b);
    if (rc) return rc;
    if (!d_hist288) return fail(FPNG_AMD_ERR_INVALID_ARG, "null histogram");
    HIP_TRY(hipSetDevice(e->device));
    if ((rc = drain(e)) || (rc = e->h_jobs.ensure(4)) || (rc = e->sc[0].d_jobs.ensure(4)) || (rc = band_record_free(e, 2))) return rc;
    Job &j = e->h_jobs.p[2];
    band_job(e, b, true, j);
    j.table = g_dev[e->device].symbols[b->num_chans];
    hipStream_t s = e->stream;
    HIP_TRY(hipMemcpyAsync(e->sc[0].d_jobs.p + 2, &j, sizeof(Job), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(e->band_copied[2], s));
    HIP_TRY(hipMemsetAsync(d_hist288, 0, 288 * sizeof(uint32_t), s));
    launch_hist(s, e->sc[0].d_jobs.p + 2, 1, j.nrows, d_hist288);
    HIP_TRY(hipGetLastError());
    return FPNG_AMD_OK;
}

int fpng_amd_band_encode(fpng_amd_encoder *e, const fpng_amd_band *b, uint32_t flags, const uint32_t *d_hist288,
                         fpng_amd_band_stats *stats)
{
    int rc = band_check(e, b);
    if (rc) return rc;
    if (!stats) return fail(FPNG_AMD_ERR_INVALID_ARG, "null stats");
    if (flags & FPNG_AMD_FORCE_UNCOMPRESSED) return fail(FPNG_AMD_ERR_INVALID_ARG, "stored images have no band path: encode them whole");
    const bool two_pass = (flags & FPNG_AMD_ENCODE_SLOWER) != 0;
    if (two_pass && !d_hist288) return fail(FPNG_AMD_ERR_INVALID_ARG, "2-pass bands need the image's histogram");
    HIP_TRY(hipSetDevice(e->device));
    fpng_amd_encoder::Scratch &sc = e->sc[0];
    if ((rc = drain(e)) || (rc = e->h_jobs.ensure(4)) || (rc = sc.d_jobs.ensure(4)) || (rc = e->h_states.ensure(2))) return rc;
    if (two_pass && (rc = sc.d_dyn.ensure(1))) return rc;
    if ((rc = band_record_free(e, 0))) return rc;
    Job &j = e->h_jobs.p[0];
    band_job(e, b, two_pass, j);
    if ((rc = sc.d_rows.ensure(j.nrows)) || (rc = sc.d_row_off.ensure(j.nrows)) || (rc = sc.d_states.ensure(1)) ||
        (rc = sc.d_local.ensure(kLocalFrontPad + (uint64_t)j.local_stride * j.nrows + 16)))
        return rc;
    hipStream_t s = e->stream;
    if (two_pass) { // the table every rank builds from the same (all-reduced) histogram
        Job &jb = e->h_jobs.p[1];
        jb = j;
        jb.table = g_dev[e->device].symbols[b->num_chans];
        HIP_TRY(hipMemcpyAsync(sc.d_jobs.p + 1, &jb, sizeof(Job), hipMemcpyHostToDevice, s));
        launch_build_dynamic(s, sc.d_jobs.p + 1, 1, d_hist288, sc.d_dyn.p, 0);
    }
    HIP_TRY(hipMemcpyAsync(sc.d_jobs.p, &j, sizeof(Job), hipMemcpyHostToDevice, s));
    launch_encode_rows(s, sc.d_jobs.p, 1, j.nrows, b->num_chans == 3 ? 1u : 2u, sc.d_rows.p, sc.d_states.p, sc.d_local.p, b->w >= kWideRowPixels);
    launch_scan(s, sc.d_jobs.p, 1, sc.d_rows.p, sc.d_row_off.p, sc.d_states.p); // band count: sums only
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(e->h_states.p, sc.d_states.p, sizeof(JobState), hipMemcpyDeviceToHost, s));
    uint32_t *ftb = (uint32_t *)(e->h_states.p + 1); // [0] first_token_bit, [1] end-of-block entry of the table in use
    if (two_pass) {
        HIP_TRY(hipMemcpyAsync(ftb, &sc.d_dyn.p->first_token_bit, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(ftb + 1, &sc.d_dyn.p->lit[256], sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    } else {
        ftb[0] = g_host_1pass[b->num_chans].first_token_bit;
        ftb[1] = g_host_1pass[b->num_chans].lit[256];
    }
    HIP_TRY(hipStreamSynchronize(s));
    const JobState &st = e->h_states.p[0];
    stats->token_bits = st.token_end_bit - (j.is_first ? ftb[0] : 0); // (scan starts the first band at the table's first token bit)
    stats->adler_s1 = st.s1;
    stats->adler_s2 = st.s2;
    stats->adler_len = (uint64_t)(j.bpl + 1) * j.nrows;
    stats->last_unit_bits = st.last_unit_bits;
    stats->first_token_bit = ftb[0];
    stats->eob_bits = ftb[1] >> 16;
    stats->reserved = 0;
    e->band_token_bits = stats->token_bits;
    e->band_eob_bits = stats->eob_bits;
    e->band_two_pass = two_pass;
    return FPNG_AMD_OK;
}

int fpng_amd_band_place(fpng_amd_encoder *e, const fpng_amd_band *b, uint64_t start_bit, uint64_t zlib_size, uint8_t *d_window,
                        size_t window_cap, uint64_t *window_file_offset, size_t *window_bytes)
{
    int rc = band_check(e, b);
    if (rc) return rc;
    if (!d_window || !window_file_offset || !window_bytes || (zlib_size && zlib_size < 6)) return fail(FPNG_AMD_ERR_INVALID_ARG, "bad argument");
    if ((uintptr_t)d_window & 15) return fail(FPNG_AMD_ERR_INVALID_ARG, "d_window must be 16-byte aligned");
    HIP_TRY(hipSetDevice(e->device));
    fpng_amd_encoder::Scratch &sc = e->sc[0];
    if ((rc = drain(e)) || (rc = e->h_jobs.ensure(4)) || (rc = sc.d_jobs.ensure(4)) || (rc = band_record_free(e, 3))) return rc;
    Job &j = e->h_jobs.p[3];
    band_job(e, b, e->band_two_pass, j);
    // the window: whole 16-byte pieces of the FILE from the piece holding the band's first bit to the one holding its last
    const uint64_t file_bit0 = (uint64_t)kPngHeaderBytes * 8 + start_bit;
    const uint32_t eob_bits = j.is_last ? e->band_eob_bits : 0u;
    const uint64_t file_bit1 = file_bit0 + e->band_token_bits + eob_bits;
    const uint64_t wb0 = j.is_first ? 0 : ((file_bit0 >> 3) & ~15ull);
    const uint64_t wb1 = (((file_bit1 + 7) >> 3) + 15) & ~15ull;
    *window_file_offset = wb0;
    *window_bytes = (size_t)(wb1 - wb0);
    if (window_cap < wb1 - wb0) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "band window too small");
    // zlib_size == 0 ("not known yet"): the band pretends that the data ends with its window; its CRC partials then
    // describe [58, wb1) with foreign bits zero, and fpng_amd_band_crc() folds them to one value positioned at wb1
    e->band_self_end = zlib_size ? 0 : wb1;
    if (!zlib_size) zlib_size = wb1 - kPngHeaderBytes + 4;
    j.start_bit = start_bit;
    j.flags = 0x100u | 0x200u; // band placement through scan_kernel + assemble_kernel
    j.band_zlib_size = zlib_size;
    j.out = d_window - wb0;    // file byte 0 as the window sees it (only bytes >= wb0 are ever touched)
    j.out_cap = window_cap + wb0;
    j.crc_blocks = (uint32_t)((kPngHeaderBytes + zlib_size + kCrcRangeBytes - 1) / kCrcRangeBytes) + 1;
    if ((rc = sc.d_partials.ensure(j.crc_blocks))) return rc;
    hipStream_t s = e->stream;
    HIP_TRY(hipMemcpyAsync(sc.d_jobs.p + 3, &j, sizeof(Job), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(e->band_copied[3], s));
    // ranges the band does not touch contribute nothing to the CRC (fpng_amd_band_crc_partials)
    HIP_TRY(hipMemsetAsync(sc.d_partials.p, 0, (size_t)j.crc_blocks * sizeof(uint32_t), s));
    launch_scan(s, sc.d_jobs.p + 3, 1, sc.d_rows.p, sc.d_row_off.p, sc.d_states.p); // absolute row offsets, stream head
    launch_assemble(s, sc.d_jobs.p + 3, 1, j.crc_blocks, sc.d_states.p, sc.d_row_off.p, sc.d_local.p, g_dev[e->device].crc,
                    sc.d_partials.p, nullptr);
    HIP_TRY(hipGetLastError());
    e->band_crc_ranges = crc_ranges_of(zlib_size);
    return FPNG_AMD_OK;
}

int fpng_amd_band_crc_partials(fpng_amd_encoder *e, uint32_t *d_partials, uint32_t cap, uint32_t *n_partials)
{
    if (!e || !n_partials) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    if (!e->band_crc_ranges) return fail(FPNG_AMD_ERR_INVALID_ARG, "no fpng_amd_band_place() before");
    *n_partials = e->band_crc_ranges;
    if (!d_partials) return FPNG_AMD_OK; // (size query)
    if (cap < e->band_crc_ranges) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "partials buffer too small");
    HIP_TRY(hipSetDevice(e->device));
    int rc = drain(e);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_partials, e->sc[0].d_partials.p, (size_t)e->band_crc_ranges * sizeof(uint32_t), hipMemcpyDeviceToDevice, e->stream));
    return FPNG_AMD_OK;
}

static int wrap_png(fpng_amd_encoder *e, uint8_t *d_png, size_t zlib_size, uint32_t adler, uint32_t w, uint32_t h, uint32_t c,
                    const uint32_t *d_crc_partials, uint32_t n_partials, size_t *png_size)
{
    if (!e || !d_png || !png_size || zlib_size < 6) return fail(FPNG_AMD_ERR_INVALID_ARG, "bad argument");
    if (d_crc_partials && n_partials != crc_ranges_of(zlib_size)) return fail(FPNG_AMD_ERR_INVALID_ARG, "wrong number of CRC partials for this stream size");
    if ((uintptr_t)d_png & 15) return fail(FPNG_AMD_ERR_INVALID_ARG, "d_png must be 16-byte aligned");
    int rc = check_dims(w, h, c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    fpng_amd_encoder::Scratch &sc = e->sc[0]; // (lane 0's scratch: whole-image submissions in flight must be done with it)
    if ((rc = drain(e)) || (rc = e->h_jobs.ensure(4)) || (rc = sc.d_jobs.ensure(4)) || (rc = sc.d_states.ensure(1)) || (rc = e->h_states.ensure(2)) ||
        (rc = sc.d_results.ensure(1)) || (rc = sc.d_rows.ensure(1)))
        return rc;
    if ((rc = band_record_free(e, 0))) return rc;
    Job &j = e->h_jobs.p[0];
    std::memset(&j, 0, sizeof j);
    j.out = d_png;
    j.w = w, j.c = c, j.bpl = w * c, j.nrows = 0, j.h_total = h;
    j.whole_png = j.is_first = j.is_last = 1;
    j.crc_blocks = (uint32_t)((kPngHeaderBytes + zlib_size + kCrcRangeBytes - 1) / kCrcRangeBytes) + 1;
    make_png_header(j.png_header, w, h, c);
    j.png_header[50] = (uint8_t)(zlib_size >> 24), j.png_header[51] = (uint8_t)(zlib_size >> 16);
    j.png_header[52] = (uint8_t)(zlib_size >> 8), j.png_header[53] = (uint8_t)zlib_size;
    JobState &st = e->h_states.p[0];
    std::memset(&st, 0, sizeof st);
    st.zlib_size = zlib_size;
    st.mode = 0;
    st.adler = adler; // finalize_kernel writes it behind the stream (reference fpng.cpp:1569-1577)
    if ((rc = sc.d_partials.ensure(j.crc_blocks))) return rc;
    hipStream_t s = e->stream;
    HIP_TRY(hipMemcpyAsync(sc.d_jobs.p, &j, sizeof(Job), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(sc.d_states.p, &st, sizeof(JobState), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_png, j.png_header, kPngHeaderBytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(e->band_copied[0], s));
    if (d_crc_partials) // the bands' partials, already XOR-ed together: no pass over the file
        HIP_TRY(hipMemcpyAsync(sc.d_partials.p, d_crc_partials, (size_t)n_partials * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    else
        launch_crc(s, sc.d_jobs.p, 1, j.crc_blocks, sc.d_states.p, g_dev[e->device].crc, sc.d_partials.p);
    launch_finalize(s, sc.d_jobs.p, 1, j.crc_blocks, sc.d_rows.p, sc.d_states.p, g_dev[e->device].crc, sc.d_partials.p, nullptr, sc.d_results.p);
    HIP_TRY(hipGetLastError());
    *png_size = kPngHeaderBytes + zlib_size + kPngTrailerBytes; // asynchronous: complete when the encoder's stream gets there
    return FPNG_AMD_OK;
}

int fpng_amd_wrap_png(fpng_amd_encoder *e, uint8_t *d_png, size_t zlib_size, uint32_t adler, uint32_t w, uint32_t h, uint32_t c,
                      size_t *png_size)
{
    return wrap_png(e, d_png, zlib_size, adler, w, h, c, nullptr, 0, png_size);
}

int fpng_amd_wrap_png_crc(fpng_amd_encoder *e, uint8_t *d_png, size_t zlib_size, uint32_t adler, uint32_t w, uint32_t h, uint32_t c,
                          const uint32_t *d_crc_partials, uint32_t n_partials, size_t *png_size)
{
    if (!d_crc_partials) return fail(FPNG_AMD_ERR_INVALID_ARG, "null partials");
    return wrap_png(e, d_png, zlib_size, adler, w, h, c, d_crc_partials, n_partials, png_size);
}

int fpng_amd_train_tables(fpng_amd_encoder *e, const fpng_amd_image *images, uint32_t n, uint32_t c, uint8_t *prefix, size_t prefix_cap,
                          size_t *prefix_bytes, uint32_t *bit_buf, uint32_t *bit_buf_size, uint32_t codes[288], uint8_t code_sizes[288])
{
    if (!e || !images || !n || !prefix || !prefix_bytes || !bit_buf || !bit_buf_size || !codes || !code_sizes)
        return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    if (c != 3 && c != 4) return fail(FPNG_AMD_ERR_INVALID_ARG, "num_chans must be 3 or 4");
    if (n > 65535) return fail(FPNG_AMD_ERR_INVALID_ARG, "corpus larger than 65535 images (the sums are 32-bit in the reference)");
    HIP_TRY(hipSetDevice(e->device));
    int rc = drain(e);
    if (rc) return rc;
    fpng_amd_encoder::Scratch &sc = e->sc[0];
    const DeviceTables &dt = g_dev[e->device];
    // jobs: the images under the symbol table of the histogram pass (no output buffers)
    PinnedBuf<Job> &hj = e->slots[0].jobs;
    if ((rc = hj.ensure(n + 1)) || (rc = sc.d_jobs.ensure(n + 1)) || (rc = sc.d_hist.ensure(((size_t)n + 3) * 288)) || (rc = sc.d_dyn.ensure(1)) ||
        (rc = e->h_states.ensure(16)))
        return rc;
    uint32_t max_rows = 0;
    for (uint32_t i = 0; i < n; i++) {
        const fpng_amd_image &im = images[i];
        if ((rc = check_dims(im.w, im.h, im.num_chans))) return rc;
        if (im.num_chans != c) return fail(FPNG_AMD_ERR_INVALID_ARG, "every image of the corpus must have num_chans channels");
        if (!im.d_pixels || (c == 4 && ((uintptr_t)im.d_pixels & 3))) return fail(FPNG_AMD_ERR_INVALID_ARG, "null / misaligned pixel pointer");
        Job &j = hj.p[i];
        std::memset(&j, 0, sizeof j);
        j.rows = (const uint8_t *)im.d_pixels;
        j.w = im.w, j.c = c, j.bpl = im.w * c, j.nrows = j.h_total = im.h;
        j.flags = FPNG_AMD_ENCODE_SLOWER;
        j.whole_png = j.is_first = j.is_last = 1;
        j.table = dt.symbols[c];
        max_rows = std::max(max_rows, im.h);
    }
    Job &jt = hj.p[n]; // the builder's job: corpus sums in, table out
    std::memset(&jt, 0, sizeof jt);
    jt.c = c, jt.flags = FPNG_AMD_ENCODE_SLOWER | 0x400u, jt.table = dt.symbols[c];
    hipStream_t s = e->stream;
    sc.hist_zero = 0;
    uint32_t *d_hist = sc.d_hist.p;                       // n x 288 counters
    uint64_t *d_sums = (uint64_t *)(d_hist + (size_t)n * 288); // 288 x u64 (n * 288 * 4 is a multiple of 8)
    uint32_t *d_freq = d_hist + (size_t)n * 288 + 576;    // the corpus histogram the builder reads
    HIP_TRY(hipMemcpyAsync(sc.d_jobs.p, hj.p, (n + 1) * sizeof(Job), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(d_hist, 0, ((size_t)n + 3) * 288 * sizeof(uint32_t), s));
    launch_hist(s, sc.d_jobs.p, n, max_rows, d_hist);
    launch_train_accumulate(s, d_hist, n, d_sums);
    HIP_TRY(hipGetLastError());
    std::vector<uint64_t> sums(288);
    HIP_TRY(hipMemcpyAsync(sums.data(), d_sums, 288 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // every literal, the end-of-block symbol and every length a multiple-of-c match can have must be codable
    // (reference fpng.cpp:941-952)
    std::vector<uint32_t> freq(288);
    for (int i = 0; i < 288; i++) freq[i] = (uint32_t)sums[i];
    for (int i = 0; i <= 256; i++)
        if (!freq[i]) freq[i] = 1;
    for (uint32_t len = c; len <= 258; len += c) {
        uint32_t sym, extra;
        deflate_length_symbol(len - 3, &sym, &extra);
        if (!freq[sym]) freq[sym] = 1;
    }
    HIP_TRY(hipMemcpyAsync(d_freq, freq.data(), 288 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    launch_build_dynamic(s, sc.d_jobs.p + n, 1, d_freq, sc.d_dyn.p, 0);
    HIP_TRY(hipGetLastError());
    std::vector<TokenTable> tab(1);
    HIP_TRY(hipMemcpyAsync(tab.data(), sc.d_dyn.p, sizeof(TokenTable), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint32_t bits = tab[0].header_bits, nbytes = bits / 8;
    *prefix_bytes = nbytes;
    if (prefix_cap < nbytes) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "prefix buffer too small");
    std::memcpy(prefix, tab[0].header, nbytes);
    *bit_buf_size = bits % 8;
    *bit_buf = tab[0].header[nbytes] & ((1u << (bits % 8)) - 1u);
    for (int i = 0; i < 288; i++) codes[i] = lit_code(tab[0].lit[i]), code_sizes[i] = (uint8_t)lit_len(tab[0].lit[i]);
    return FPNG_AMD_OK;
}

int fpng_amd_debug_peek(fpng_amd_encoder *e, int lane, uint32_t *dst, uint32_t n_words)
{
    if (!e || lane < 0 || lane >= fpng_amd_encoder::kLanes || !dst) return fail(FPNG_AMD_ERR_INVALID_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    if (n_words == 8) { // the timing build's cycle counts of build_dynamic_kernel (head of the histogram scratch)
        if (!e->sc[lane].d_hist.p) return fail(FPNG_AMD_ERR_INVALID_ARG, "no 2-pass submission yet");
        const uint32_t page = (dst[7] == 0xFEEDu) ? 8u : 0u; // (second page: the table builder's inner phases)
        HIP_TRY(hipMemcpy(dst, e->sc[lane].d_hist.p + page, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return FPNG_AMD_OK;
    }
    return fail(FPNG_AMD_ERR_INVALID_ARG, "unknown debug page");
}

} // extern "C"
