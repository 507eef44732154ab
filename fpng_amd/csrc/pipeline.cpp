// pipeline.cpp -- host-side pipelines on top of the device-resident encode path (api.cpp):
//
//   * the arithmetic of the row-band path -- start bits, Adler-32 combine, the reference's compressed-or-stored rule, the
//     IDAT CRC-32 from per-band raw CRCs, PNG head and tail -- as plain host functions of the C ABI (no GPU), shared by the
//     streamed host path below and by sharded.cpp (one image over several GPUs);
//   * fpng_amd_encode_host_to(): fpng_encode_image_to_memory() on HOST buffers (reference src/fpng.cpp:1662-1803) with the
//     frame streamed through the GPU in row bands: upload of band k+1 | encode + placement of band k | download of band k-1;
//   * fpng_amd_node_*: one process, every GPU of the node, frames dealt round-robin.
//
// No pixel is touched on the CPU here, and there is no CPU encoder: without a GPU every encode entry point fails.
#include "encoder.h"
#include "host_workers.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

using namespace fpng_amd;

extern "C" {

// ------------------------------------------------------------------------------------------------
// band arithmetic (the Python mirror is fpng_amd/sharded.py: plan_bands / window_extent)
// ------------------------------------------------------------------------------------------------
int fpng_amd_plan_bands(const fpng_amd_band_stats *stats, uint32_t n, uint32_t w, uint32_t h, uint32_t c, uint32_t flags,
                        uint64_t *start_bits, fpng_amd_band_plan *plan)
{
    if (!stats || !n || !plan) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    int rc = check_dims(w, h, c);
    if (rc) return rc;
    const bool one_pass = !(flags & FPNG_AMD_ENCODE_SLOWER);
    uint32_t first = n, last_unit = 0;
    for (uint32_t k = 0; k < n; k++)
        if (stats[k].adler_len) {
            if (first == n) first = k;
            last_unit = stats[k].last_unit_bits;
        }
    if (first == n) return fail(FPNG_AMD_ERR_INVALID_ARG, "no band has rows");
    const uint32_t first_token_bit = stats[first].first_token_bit, eob_bits = stats[first].eob_bits;
    uint64_t pos = first_token_bit;
    // Adler-32 of X||Y from raw sums: S1 adds, S2(XY) = S2(X) + |Y| * S1(X) + S2(Y)
    uint64_t s1 = 0, s2 = 0, nbytes = 0;
    for (uint32_t k = 0; k < n; k++) {
        if (start_bits) start_bits[k] = pos;
        pos += stats[k].token_bits;
        s2 = (s2 + (stats[k].adler_len % kAdlerMod) * s1 + stats[k].adler_s2) % kAdlerMod; // s1 = S1 of everything before
        s1 = (s1 + stats[k].adler_s1) % kAdlerMod;
        nbytes += stats[k].adler_len;
    }
    const uint64_t n_total = ((uint64_t)w * c + 1) * h;
    if (nbytes != n_total) return fail(FPNG_AMD_ERR_INVALID_ARG, "the bands do not cover the image");
    plan->end_bit = pos;
    plan->adler = (uint32_t)((((nbytes % kAdlerMod) + s2) % kAdlerMod) << 16) | (uint32_t)((1 + s1) % kAdlerMod);
    // the reference's bit writer gives up when it gets within 8 bytes of its buffer (src/fpng.cpp:567-588), closed form
    const uint64_t D = ((58 + n_total + 7) & ~7ull) - 58;
    plan->stored = ((one_pass && D < first_token_bit / 8) || (((pos - last_unit) >> 3) + 8 > D) || (((pos + eob_bits + 7) >> 3) + 4 > D)) ? 1u : 0u;
    plan->zlib_size = ((pos + eob_bits + 7) >> 3) + 4;
    return FPNG_AMD_OK;
}

int fpng_amd_band_window(int is_first, int is_last, uint64_t start_bit, uint64_t token_bits, uint32_t eob_bits, uint64_t *file_offset,
                         size_t *bytes, uint32_t *shared_head)
{
    const uint64_t fb0 = (uint64_t)kPngHeaderBytes * 8 + start_bit, fb1 = fb0 + token_bits + (is_last ? eob_bits : 0u);
    const uint64_t wb0 = is_first ? 0 : ((fb0 >> 3) & ~15ull), wb1 = (((fb1 + 7) >> 3) + 15) & ~15ull;
    if (file_offset) *file_offset = wb0;
    if (bytes) *bytes = (size_t)(wb1 - wb0);
    if (shared_head) *shared_head = (!is_first && (fb0 & 127)) ? (uint32_t)std::min<uint64_t>(16, wb1 - wb0) : 0u;
    return FPNG_AMD_OK;
}

uint32_t fpng_amd_idat_crc_from_bands(const uint32_t *raw, const uint64_t *end_offset, uint32_t n, uint64_t zlib_size, uint32_t adler)
{
    // every band's value is the raw CRC of the file's bytes [58, end_k) with all other bands' bits zero; a raw CRC is linear,
    // so moved to the common end point (the first Adler byte) they simply XOR together
    const uint64_t data_end = kPngHeaderBytes + zlib_size - 4, ord = 0xFFFFFFFFull;
    uint32_t data = 0;
    for (uint32_t k = 0; k < n; k++) {
        if (!raw[k]) continue;
        const uint64_t e = end_offset[k];
        // (a window may end up to 15 zero bytes behind the data: a negative shift, x has order dividing 2^32-1)
        const uint32_t f = e <= data_end ? gf2_xpow8n(data_end - e) : gf2_xpow((ord - (8 * (e - data_end)) % ord) % ord);
        data ^= gf2_mulmod(raw[k], f);
    }
    uint32_t s = ~host_crc32("IDAT", 4, 0); // running state (init ~0) after the chunk type
    s = gf2_mulmod(s, gf2_xpow8n(zlib_size - 4)) ^ data;
    const uint8_t a[4] = {(uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler};
    return host_crc32(a, 4, ~s);
}

int fpng_amd_png_head(uint32_t w, uint32_t h, uint32_t c, uint64_t zlib_size, uint8_t head[58])
{
    int rc = check_dims(w, h, c);
    if (rc) return rc;
    uint8_t hdr[60];
    make_png_header(hdr, w, h, c);
    hdr[50] = (uint8_t)(zlib_size >> 24), hdr[51] = (uint8_t)(zlib_size >> 16), hdr[52] = (uint8_t)(zlib_size >> 8), hdr[53] = (uint8_t)zlib_size;
    std::memcpy(head, hdr, kPngHeaderBytes);
    return FPNG_AMD_OK;
}

void fpng_amd_png_tail(uint32_t adler, uint32_t crc, uint8_t tail[20])
{
    static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
    tail[0] = (uint8_t)(adler >> 24), tail[1] = (uint8_t)(adler >> 16), tail[2] = (uint8_t)(adler >> 8), tail[3] = (uint8_t)adler;
    tail[4] = (uint8_t)(crc >> 24), tail[5] = (uint8_t)(crc >> 16), tail[6] = (uint8_t)(crc >> 8), tail[7] = (uint8_t)crc;
    std::memcpy(tail + 8, iend, 12);
}

} // extern "C"

namespace {

// partial j of a band placed with zlib_size == 0 covers the 64 KiB range that ends j ranges before the window's end and is
// positioned one block row (kCrcRowBytes) behind its range (assemble_kernel / finalize_kernel): fold to the window's end
uint32_t fold_partials(const uint32_t *p, uint32_t n)
{
    // Horner with X = x^(8 * 64 KiB); multiplication by a constant is linear over GF(2): four byte-indexed tables (a 16384^2
    // image has 7000 partials: the bitwise product would cost 0.7 ms)
    static uint32_t mulX[4][256];
    static std::once_flag once;
    std::call_once(once, [] {
        const uint32_t X = gf2_xpow8n(kCrcRangeBytes);
        for (uint32_t k = 0; k < 4; k++)
            for (uint32_t b = 0; b < 256; b++) mulX[k][b] = gf2_mulmod(b << (8 * k), X);
    });
    uint32_t acc = 0;
    for (uint32_t j = n; j-- > 0;) // partial n-1 is the oldest
        acc = mulX[0][acc & 0xFF] ^ mulX[1][(acc >> 8) & 0xFF] ^ mulX[2][(acc >> 16) & 0xFF] ^ mulX[3][acc >> 24] ^ p[j];
    const uint64_t ord = 0xFFFFFFFFull;
    return gf2_mulmod(acc, gf2_xpow((ord - (8ull * kCrcRowBytes) % ord) % ord));
}

uint32_t crc_ranges_for_end(uint64_t end_aligned) { return (uint32_t)((end_aligned - 48 + kCrcRangeBytes - 1) / kCrcRangeBytes); }

} // namespace

extern "C" int fpng_amd_band_crc(fpng_amd_encoder *e, uint32_t *raw_crc, uint64_t *end_offset)
{
    if (!e || !raw_crc || !end_offset) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    if (!e->band_self_end || !e->band_crc_ranges) return fail(FPNG_AMD_ERR_INVALID_ARG, "no fpng_amd_band_place() with zlib_size == 0 before");
    HIP_TRY(hipSetDevice(e->device));
    int rc = e->h_partials.ensure(e->band_crc_ranges);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(e->h_partials.p, e->sc[0].d_partials.p, (size_t)e->band_crc_ranges * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    *raw_crc = fold_partials(e->h_partials.p, e->band_crc_ranges);
    *end_offset = e->band_self_end;
    return FPNG_AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// fpng_amd_encode_host_to(): host pixels in, host PNG out
// ------------------------------------------------------------------------------------------------
void fpng_amd::destroy_host_workers(fpng_amd_encoder *e)
{
    delete e->workers;
    e->workers = nullptr;
}

// The host paths' two copy streams, made so that the two PCIe directions run on DIFFERENT copy engines.
//
// What the runtime does (its own log, AMD_LOG_LEVEL=4, "HSA Copy copy_engine=0x1 ... engineType=2", tools/gpu_host_state_log.sh):
// a stream's FIRST copy asks which SDMA engines are idle at that instant, takes the lowest one and keeps it for the stream's
// life.  The upload stream takes engine 1.  The download stream's first copy used to be issued by the downloader thread whenever
// the first frame or band was encoded -- if that fell between two of the uploader's 32 MiB chunks (the runtime pins pageable
// memory chunk by chunk; ~80 us of idle engine between chunks), it took engine 1 as well, and from then on every download of
// the process queued behind the uploads: 3.43-3.58 ms per 8K frame instead of 2.59-2.85, in roughly one process in ten on one
// box and in every process on another (profiles/r03_host_path.txt).  So the first download is issued here, from this thread,
// right behind an 8 MiB upload that is still in flight (pinned source: the call returns at once and the engine is busy for
// ~150 us), and the outcome is measured: both directions together must take clearly less than one after the other; if not the
// download stream is made anew and the step repeated.
//
// (A second, unrelated cliff looked like this one for a while: device memory that was hipFree'd and allocated again downloads at
// half speed -- see DeviceBuf in encoder.h.)
int fpng_amd::ensure_copy_streams(fpng_amd_encoder *e)
{
    auto &ring = e->host;
    if (ring.up && ring.down) return FPNG_AMD_OK;
    static const bool trace = getenv("FPNG_AMD_TRACE") != nullptr;
    constexpr size_t kBytes = 8u << 20;
    PinnedBuf<uint8_t> h;
    DeviceBuf<uint8_t> d;
    int rc;
    if ((rc = h.ensure(2 * kBytes)) || (rc = d.ensure(2 * kBytes))) return rc;
    struct Free {
        PinnedBuf<uint8_t> &h;
        DeviceBuf<uint8_t> &d;
        ~Free() { h.release(), d.release(); }
    } free_them{h, d};
    memset(h.p, 0, kBytes);
    if (!ring.up) HIP_TRY(create_copy_stream(&ring.up));
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); };
    for (int attempt = 0; attempt < 4; attempt++) {
        if (!ring.down) HIP_TRY(create_copy_stream(&ring.down));
        double t_up = 1e30, t_down = 1e30, t_both = 1e30;
        for (int rep = 0; rep < 2; rep++) {
            // (the very first pass through here is the one that settles the download stream's engine: upload in flight, download behind it)
            auto t0 = clk::now();
            HIP_TRY(hipMemcpyAsync(d.p, h.p, kBytes, hipMemcpyHostToDevice, ring.up));
            HIP_TRY(hipMemcpyAsync(h.p + kBytes, d.p + kBytes, kBytes, hipMemcpyDeviceToHost, ring.down));
            HIP_TRY(hipStreamSynchronize(ring.up));
            HIP_TRY(hipStreamSynchronize(ring.down));
            t_both = std::min(t_both, us(t0));
            t0 = clk::now();
            HIP_TRY(hipMemcpyAsync(d.p, h.p, kBytes, hipMemcpyHostToDevice, ring.up));
            HIP_TRY(hipStreamSynchronize(ring.up));
            t_up = std::min(t_up, us(t0));
            t0 = clk::now();
            HIP_TRY(hipMemcpyAsync(h.p + kBytes, d.p + kBytes, kBytes, hipMemcpyDeviceToHost, ring.down));
            HIP_TRY(hipStreamSynchronize(ring.down));
            t_down = std::min(t_down, us(t0));
        }
        const bool apart = t_both < 0.8 * (t_up + t_down);
        if (trace)
            fprintf(stderr, "fpng_amd: copy streams, attempt %d: up %.0f us, down %.0f us, both %.0f us -> %s\n", attempt, t_up, t_down, t_both,
                    apart ? "two engines" : "one engine, again");
        if (apart) break;
        if (attempt == 3) break; // (keep the last pair: the path is correct either way, only slower)
        (void)hipStreamDestroy(ring.down);
        ring.down = nullptr;
    }
    return FPNG_AMD_OK;
}
namespace {

struct FixedOut {
    uint8_t *p;
    size_t cap;
};
uint8_t *fixed_reserve(void *user, size_t bytes)
{
    FixedOut *f = (FixedOut *)user;
    return bytes <= f->cap ? f->p : nullptr;
}

// upload everything, one whole-image submission, fetch the size, download: 2-pass, forced-stored and small frames, and the
// way out when a streamed frame turns out incompressible (`uploaded`: the pixels are on the device already)
int encode_host_serial(fpng_amd_encoder *e, const void *pixels, bool uploaded, uint32_t w, uint32_t h, uint32_t c, uint32_t flags,
                       fpng_amd_reserve_fn reserve, void *user, size_t *out_size)
{
    const size_t in_bytes = (size_t)w * h * c, max_out = fpng_amd_max_encoded_size(w, h, c);
    int rc;
    if ((rc = e->d_stage_in.ensure(in_bytes + 16)) || (rc = e->d_stage_out.ensure(max_out + 64))) return rc;
    if (!uploaded) HIP_TRY(hipMemcpyAsync(e->d_stage_in.p, pixels, in_bytes, hipMemcpyHostToDevice, e->stream));
    fpng_amd_image im;
    im.d_pixels = e->d_stage_in.p;
    im.w = w, im.h = h, im.num_chans = c;
    im.d_out = e->d_stage_out.p;
    im.out_cap = e->d_stage_out.cap;
    if ((rc = fpng_amd_encode_batch_async(e, &im, 1, flags))) return rc;
    fpng_amd_result res;
    if ((rc = fpng_amd_encode_finish(e, &res, 1))) return rc;
    if (res.status) return fail(FPNG_AMD_ERR_HIP, "device reported an encode failure");
    *out_size = (size_t)res.png_size;
    uint8_t *out = reserve(user, (size_t)res.png_size); // the size is known before a single output byte moves
    if (!out) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "output buffer too small");
    HIP_TRY(hipMemcpy(out, e->d_stage_out.p, res.png_size, hipMemcpyDeviceToHost));
    return FPNG_AMD_OK;
}

int host_bands_forced()
{
    static const int forced = [] {
        const char *v = getenv("FPNG_AMD_HOST_BANDS");
        return v ? atoi(v) : 0;
    }();
    return forced;
}

uint32_t host_bands_for(size_t in_bytes, uint32_t h)
{
    const int forced = host_bands_forced();
    uint32_t nb = forced > 0 ? (uint32_t)forced : (uint32_t)std::min<size_t>(8, in_bytes >> 23); // ~8 MiB of pixels per band and more
    nb = std::min(nb, std::min(h, 32u));
    return std::max(nb, 1u);
}

} // namespace

extern "C" int fpng_amd_encode_host_to(fpng_amd_encoder *e, const void *pixels, uint32_t w, uint32_t h, uint32_t c, uint32_t flags,
                                       fpng_amd_reserve_fn reserve, void *user, size_t *out_size)
{
    if (!e || !pixels || !out_size || !reserve) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    int rc = check_dims(w, h, c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    const size_t bpl = (size_t)w * c, in_bytes = bpl * h, max_out = fpng_amd_max_encoded_size(w, h, c);
    // Frames of 16 MiB of pixels and more are streamed in row bands (pageable or page-locked memory alike, seen before or not:
    // tools/gpu_stream_always.sh); 2-pass and forced-stored frames need the whole image's counts first and go the serial way.
    uint32_t nb = 1;
    if (!(flags & (FPNG_AMD_ENCODE_SLOWER | FPNG_AMD_FORCE_UNCOMPRESSED))) nb = host_bands_for(in_bytes, h);
    // Band k covers rows ys[k] .. ys[k+1].  What a call costs beyond its uploads is the LAST band's walk (one row's latency
    // whatever the band's size) + placement + download, so the last of the equal bands is cut once more, 3/4 + 1/4: the
    // final download shrinks to a quarter while the 3/4 piece's upload still covers the quarter's turn on the calling thread
    // (~110 us per band).  8K RGBA: the tail after the last upload 300 -> ~210 us.
    std::vector<uint32_t> ys;
    {
        const uint32_t rows_per = (h + nb - 1) / nb;
        for (uint32_t y = 0; y < h; y += rows_per) ys.push_back(y);
        const uint32_t y0 = ys.back(), n = h - y0;
        if (nb >= 2 && !host_bands_forced() && n >= 64 && (size_t)n * bpl >= (8u << 20)) ys.push_back(y0 + n - n / 4);
        ys.push_back(h);
        nb = (uint32_t)ys.size() - 1;
    }
    e->last_host_bands = nb;
    if (nb < 2) return encode_host_serial(e, pixels, false, w, h, c, flags, reserve, user, out_size);

    // ---- streamed: band k+1 goes up while band k is encoded and placed and band k-1's window comes down ----
    if ((rc = drain(e))) return rc;
    auto &ring = e->host;
    const size_t out_cap = max_out + 64 + 32 * (size_t)nb;
    const uint32_t max_ranges = crc_ranges_for_end((kPngHeaderBytes + max_out + 15) & ~15ull) + 2;
    const size_t rec_words = 4 + (size_t)max_ranges; // per band: the window's first 16 bytes + its CRC partials
    if ((rc = e->d_stage_in.ensure(in_bytes + 16)) || (rc = e->d_stage_out.ensure(out_cap)) ||
        (rc = e->d_stream_partials.ensure((size_t)nb * rec_words)) || (rc = e->h_partials.ensure((size_t)nb * rec_words)))
        return rc;
    struct BandRun {
        uint32_t y0 = 0, y1 = 0;
        uint64_t file_off = 0;   // of the window
        size_t bytes = 0, dev_off = 0;
        uint32_t head = 0, n_part = 0;
        uint64_t self_end = 0;
        hipEvent_t placed = nullptr;
    };
    std::vector<BandRun> runs(nb);
    std::vector<fpng_amd_band_stats> stats(nb);
    std::vector<uint64_t> start_bits(nb);
    std::mutex mu;
    std::condition_variable cv;
    uint32_t uploaded = 0, placed = 0; // bands whose pixels are on the device / whose window is on its way
    bool stop = false;                 // no more bands will be placed (done, stored outcome or failure)
    std::atomic<int> failed{0};
    const int device = e->device;
    uint8_t *d_in = e->d_stage_in.p, *d_out = e->d_stage_out.p;
    // FPNG_AMD_TRACE=1: per-band timeline of the three actors on stderr (microseconds from the start of the call)
    static const bool trace = getenv("FPNG_AMD_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto now_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(); };
    std::vector<double> tl(trace ? (size_t)nb * 6 : 0, 0.0); // per band: upload begin/end, counted, placed (enqueued), download begin/end
    if ((rc = ensure_copy_streams(e))) return rc;
    for (auto &r : runs)
        if (hipEventCreateWithFlags(&r.placed, hipEventDisableTiming) != hipSuccess) failed = FPNG_AMD_ERR_HIP;
    hipStream_t s_up = ring.up, s_down = ring.down;
    // The two copy threads live as long as the encoder (no thread start per call).
    if (!e->workers) e->workers = new HostWorkers();
    HostWorkers &wk = *e->workers;
    wk.up.start([&] {
        (void)hipSetDevice(device);
        for (uint32_t k = 0; k < nb && !failed; k++) {
            const uint32_t y0 = ys[k], y1 = ys[k + 1];
            if (trace) tl[k * 6 + 0] = now_us();
            if (y1 > y0 && (hipMemcpyAsync(d_in + (size_t)y0 * bpl, (const uint8_t *)pixels + (size_t)y0 * bpl, (size_t)(y1 - y0) * bpl, hipMemcpyHostToDevice, s_up) != hipSuccess ||
                            hipStreamSynchronize(s_up) != hipSuccess))
                failed = FPNG_AMD_ERR_HIP;
            if (trace) tl[k * 6 + 1] = now_us();
            {
                std::lock_guard<std::mutex> lk(mu);
                uploaded = k + 1;
            }
            cv.notify_all();
        }
        cv.notify_all();
    });
    uint8_t *out = nullptr; // as returned by the last reserve()
    wk.down.start([&] {
        (void)hipSetDevice(device);
        for (uint32_t k = 0; k < nb; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return placed > k || stop || failed; });
                if (placed <= k) return;
            }
            const BandRun &r = runs[k];
            if (!r.bytes) continue;
            if (hipEventSynchronize(r.placed) != hipSuccess) failed = FPNG_AMD_ERR_HIP;
            if (failed) return;
            if (trace) tl[k * 6 + 4] = now_us();
            // the first request asks for an estimate of the whole file (band 0's share of the rows, plus a margin), so that
            // a growing container (std::vector) is sized once instead of band after band
            // (a hint, not a requirement: a caller's fixed buffer may be smaller than the estimate and still hold the file -- then
            //  only what this band needs is asked for)
            const size_t exact = (size_t)r.file_off + r.bytes;
            size_t want = exact;
            if (k == 0) want = std::max(want, std::min(max_out, (size_t)((double)r.bytes * h / (r.y1 - r.y0) * 1.08) + 4096));
            out = reserve(user, want);
            if (!out && want > exact) out = reserve(user, exact);
            if (!out) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    failed = FPNG_AMD_ERR_BUFFER_TOO_SMALL;
                }
                cv.notify_all();
                return;
            }
            // ONE copy per band on this stream: the band's CRC partials and the 16-byte piece it shares with its predecessor are
            // collected on the device and fetched once at the end (a few small copies per band cost more than they move).
            bool ok = true;
            if (r.bytes > r.head)
                ok = hipMemcpyAsync(out + r.file_off + r.head, d_out + r.dev_off + r.head, r.bytes - r.head, hipMemcpyDeviceToHost, s_down) == hipSuccess &&
                     hipStreamSynchronize(s_down) == hipSuccess;
            if (!ok) {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    failed = FPNG_AMD_ERR_HIP;
                }
                cv.notify_all();
                return;
            }
            if (trace) tl[k * 6 + 5] = now_us();
        }
    });

    // the calling thread: encode and place band after band as the uploads land
    bool stored = false;
    const uint64_t n_total = ((uint64_t)bpl + 1) * h, D = ((58 + n_total + 7) & ~7ull) - 58;
    size_t dev_off = 0;
    for (uint32_t k = 0; k < nb && !failed && !stored; k++) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return uploaded > k || failed; });
        }
        if (failed) break;
        BandRun &r = runs[k];
        r.y0 = ys[k], r.y1 = ys[k + 1];
        std::memset(&stats[k], 0, sizeof stats[k]);
        if (r.y1 > r.y0) {
            fpng_amd_band band;
            band.d_rows = d_in + (size_t)r.y0 * bpl;
            band.d_row_above = r.y0 ? d_in + (size_t)(r.y0 - 1) * bpl : nullptr;
            band.w = w, band.num_chans = c, band.y0 = r.y0, band.y1 = r.y1, band.h_total = h, band.reserved = 0;
            if ((rc = fpng_amd_band_encode(e, &band, 0, nullptr, &stats[k]))) { // (waits for the band's counts)
                failed = rc;
                break;
            }
            start_bits[k] = k ? start_bits[k - 1] + stats[k - 1].token_bits : stats[k].first_token_bit;
            if (trace) tl[k * 6 + 2] = now_us();
            // once the stream cannot fit the reference's buffer any more the image ends up stored (the positions only grow)
            if (((start_bits[k] + stats[k].token_bits + 7) >> 3) + 4 > D) {
                stored = true;
                break;
            }
            uint64_t fo = 0;
            size_t nbytes = 0;
            // the band's CRC partials and the first 16 bytes of its window (the piece it may share with its predecessor) are set
            // aside on the device: record k = [16 bytes of window head][partials]
            uint32_t *rec = e->d_stream_partials.p + (size_t)k * rec_words;
            if ((rc = fpng_amd_band_place(e, &band, start_bits[k], 0, d_out + dev_off, out_cap - dev_off, &fo, &nbytes)) ||
                (rc = fpng_amd_band_crc_partials(e, rec + 4, max_ranges, &r.n_part)) ||
                hipMemcpyAsync(rec, d_out + dev_off, 16, hipMemcpyDeviceToDevice, e->stream) != hipSuccess ||
                hipEventRecord(r.placed, e->stream) != hipSuccess) {
                failed = rc ? rc : FPNG_AMD_ERR_HIP;
                break;
            }
            if (trace) tl[k * 6 + 3] = now_us();
            r.file_off = fo, r.bytes = nbytes, r.dev_off = dev_off, r.self_end = e->band_self_end;
            r.head = (k && ((kPngHeaderBytes * 8 + start_bits[k]) & 127)) ? (uint32_t)std::min<size_t>(16, nbytes) : 0u;
            dev_off += (nbytes + 15) & ~(size_t)15;
        } else {
            start_bits[k] = k ? start_bits[k - 1] + stats[k - 1].token_bits : 0;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            placed = k + 1;
        }
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv.notify_all();
    wk.up.wait();
    wk.down.wait();
    for (auto &r : runs)
        if (r.placed) (void)hipEventDestroy(r.placed);
    if (trace) {
        for (uint32_t k = 0; k < nb; k++)
            fprintf(stderr, "band %2u: up %7.0f-%7.0f  counted %7.0f  placed %7.0f  down %7.0f-%7.0f us  (%zu B)\n", k, tl[k * 6], tl[k * 6 + 1],
                    tl[k * 6 + 2], tl[k * 6 + 3], tl[k * 6 + 4], tl[k * 6 + 5], runs[k].bytes);
        fprintf(stderr, "joined %7.0f us\n", now_us());
    }
    if (failed) return fail(failed, "streamed host encode failed (copy, encode or output buffer)");

    fpng_amd_band_plan plan;
    if (!stored) {
        if ((rc = fpng_amd_plan_bands(stats.data(), nb, w, h, c, 0, nullptr, &plan))) return rc;
        stored = plan.stored != 0;
    }
    if (stored) // incompressible: the reference's stored-block outcome, decided and written by the whole-image path
        return e->last_host_bands = 1, encode_host_serial(e, pixels, true, w, h, c, flags, reserve, user, out_size);

    // ---- the container around the windows (reference src/fpng.cpp:1764-1800), a few dozen bytes on the host ----
    HIP_TRY(hipMemcpyAsync(e->h_partials.p, e->d_stream_partials.p, (size_t)nb * rec_words * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::vector<uint32_t> raw(nb, 0);
    std::vector<uint64_t> ends(nb, 0);
    for (uint32_t k = 0; k < nb; k++)
        if (runs[k].bytes) {
            raw[k] = fold_partials(e->h_partials.p + (size_t)k * rec_words + 4, runs[k].n_part);
            ends[k] = runs[k].self_end;
        }
    const size_t png_size = kPngHeaderBytes + (size_t)plan.zlib_size + kPngTrailerBytes;
    out = reserve(user, png_size);
    if (!out) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "output buffer too small");
    // the pieces neighbouring windows share: each band wrote zeros where the other's bits are
    for (uint32_t k = 0; k < nb; k++) {
        const uint8_t *seam = (const uint8_t *)(e->h_partials.p + (size_t)k * rec_words);
        for (uint32_t i = 0; i < runs[k].head; i++) out[runs[k].file_off + i] |= seam[i];
    }
    uint8_t head[58], tail[20];
    if ((rc = fpng_amd_png_head(w, h, c, plan.zlib_size, head))) return rc;
    fpng_amd_png_tail(plan.adler, fpng_amd_idat_crc_from_bands(raw.data(), ends.data(), nb, plan.zlib_size, plan.adler), tail);
    std::memcpy(out, head, kPngHeaderBytes);
    std::memcpy(out + kPngHeaderBytes + plan.zlib_size - 4, tail, 20);
    *out_size = png_size;
    return FPNG_AMD_OK;
}

extern "C" int fpng_amd_encoder_last_host_bands(fpng_amd_encoder *e) { return e ? (int)e->last_host_bands : 0; }

extern "C" int fpng_amd_pin_host_memory(void *p, size_t bytes)
{
    if (!p || !bytes) return fail(FPNG_AMD_ERR_INVALID_ARG, "null/empty range");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return FPNG_AMD_OK;
}

extern "C" int fpng_amd_unpin_host_memory(void *p)
{
    if (!p) return fail(FPNG_AMD_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(hipHostUnregister(p));
    return FPNG_AMD_OK;
}

extern "C" int fpng_amd_encode_host(fpng_amd_encoder *e, const void *pixels, uint32_t w, uint32_t h, uint32_t c, uint32_t flags,
                                    uint8_t *out, size_t out_cap, size_t *out_size)
{
    if (!e || !pixels || !out_size) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    FixedOut f = {out, out ? out_cap : 0};
    return fpng_amd_encode_host_to(e, pixels, w, h, c, flags, fixed_reserve, &f, out_size);
}

// ------------------------------------------------------------------------------------------------
// whole node from one process
// ------------------------------------------------------------------------------------------------
struct fpng_amd_node {
    std::vector<fpng_amd_encoder *> enc;
};

extern "C" {

int fpng_amd_node_create(fpng_amd_node **out, const int *devices, uint32_t n)
{
    if (!out || !devices || !n) return fail(FPNG_AMD_ERR_INVALID_ARG, "null/empty device list");
    *out = nullptr;
    fpng_amd_node *node = new fpng_amd_node();
    for (uint32_t i = 0; i < n; i++) {
        fpng_amd_encoder *e = nullptr;
        int rc = fpng_amd_encoder_create(&e, devices[i], nullptr);
        if (rc) {
            fpng_amd_node_destroy(node);
            return rc;
        }
        node->enc.push_back(e);
    }
    *out = node;
    return FPNG_AMD_OK;
}

void fpng_amd_node_destroy(fpng_amd_node *node)
{
    if (!node) return;
    for (auto *e : node->enc) fpng_amd_encoder_destroy(e);
    delete node;
}

uint32_t fpng_amd_node_size(const fpng_amd_node *node) { return node ? (uint32_t)node->enc.size() : 0u; }

// ONE host-resident image over the node's devices (BASELINE config 4): contiguous row bands, one per listed device;
// every device moves ITS band up and ITS window of the file down over ITS OWN PCIe link, straight from / into the caller's
// memory -- nothing is funnelled through one GPU.  Steps 1-6 of SURVEY 8(e): bands counted (2-pass: histograms summed on the
// host first), the 64-byte records meet on the host (fpng_amd_plan_bands: start bits, Adler-32, the reference's
// stored-or-compressed rule), bands placed at their bit positions, windows downloaded to their file offsets, the pieces two
// windows share OR-ed on the host, the IDAT CRC combined from the bands' raw CRCs (fpng_amd_idat_crc_from_bands).  The output
// is byte-identical to fpng_encode_image_to_memory()'s (reference src/fpng.cpp:1662-1803): one IDAT, one Deflate block.
int fpng_amd_node_encode_host_image(fpng_amd_node *node, const void *pixels, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, fpng_amd_reserve_fn reserve,
                                    void *user, size_t *out_size)
{
    if (!node || node->enc.empty() || !pixels || !reserve || !out_size) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    int rc = check_dims(w, h, c);
    if (rc) return rc;
    const uint32_t nd = (uint32_t)node->enc.size();
    if ((flags & FPNG_AMD_FORCE_UNCOMPRESSED) || nd == 1 || h < 2 * nd) // (stored files and tiny images: one device, whole)
        return fpng_amd_encode_host_to(node->enc[0], pixels, w, h, c, flags, reserve, user, out_size);
    const bool two_pass = (flags & FPNG_AMD_ENCODE_SLOWER) != 0;
    const size_t bpl = (size_t)w * c;
    std::vector<uint32_t> ys(nd + 1);
    for (uint32_t k = 0; k <= nd; k++) ys[k] = (uint32_t)((uint64_t)h * k / nd);
    std::vector<fpng_amd_band_stats> stats(nd);
    std::vector<uint64_t> start_bits(nd);
    std::vector<uint32_t> raw(nd, 0), hist((size_t)nd * 288, 0), hist_sum(288, 0);
    std::vector<uint64_t> ends(nd, 0);
    struct Seam {
        uint8_t b[16];
        uint64_t file_off;
        uint32_t n;
    };
    std::vector<Seam> seams(nd);
    std::vector<int> rcs(nd, FPNG_AMD_OK);
    std::vector<std::string> errs(nd);
    fpng_amd_band_plan plan;
    std::memset(&plan, 0, sizeof plan);
    uint8_t *out = nullptr;
    // a barrier for the device threads; the last one to arrive runs `serial` (under the lock) before the others go on
    std::mutex mu;
    std::condition_variable cv;
    uint32_t waiting = 0, generation = 0;
    bool abort_all = false;
    auto meet = [&](const std::function<void()> &serial) {
        std::unique_lock<std::mutex> lk(mu);
        const uint32_t gen = generation;
        if (++waiting == nd) {
            if (!abort_all)
                for (int v : rcs)
                    if (v) abort_all = true;
            if (!abort_all && serial) serial();
            waiting = 0, generation++;
            cv.notify_all();
        } else
            cv.wait(lk, [&] { return generation != gen; });
        return !abort_all;
    };
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < nd; d++)
        th.emplace_back([&, d] {
            fpng_amd_encoder *e = node->enc[d];
            const uint32_t y0 = ys[d], y1 = ys[d + 1], rows = y1 - y0;
            auto bail = [&](int code) {
                rcs[d] = code;
                errs[d] = fpng_amd_last_error();
            };
            uint32_t *d_hist = nullptr;
            fpng_amd_band band;
            std::memset(&band, 0, sizeof band);
            do { // ---- phase A: the band goes up (with the row above it), is counted ----
                if (hipSetDevice(e->device) != hipSuccess || drain(e)) { bail(FPNG_AMD_ERR_HIP); break; }
                const size_t up_rows = rows + (y0 ? 1 : 0);
                int r2;
                if ((r2 = e->d_stage_in.ensure(up_rows * bpl + 16 + 288 * 4)) || (r2 = e->d_stage_out.ensure(fpng_amd_max_encoded_size(w, std::max(rows, 1u), c) + 256))) { bail(r2); break; }
                if (rows && hipMemcpyAsync(e->d_stage_in.p, (const uint8_t *)pixels + (size_t)(y0 - (y0 ? 1 : 0)) * bpl, up_rows * bpl, hipMemcpyHostToDevice, e->stream) != hipSuccess) { bail(FPNG_AMD_ERR_HIP); break; }
                d_hist = (uint32_t *)(e->d_stage_in.p + ((up_rows * bpl + 15) & ~(size_t)15));
                band.d_rows = e->d_stage_in.p + (y0 ? bpl : 0), band.d_row_above = y0 ? e->d_stage_in.p : nullptr;
                band.w = w, band.num_chans = c, band.y0 = y0, band.y1 = y1, band.h_total = h;
                if (two_pass && rows) {
                    if ((r2 = fpng_amd_band_hist(e, &band, d_hist))) { bail(r2); break; }
                    if (hipMemcpyAsync(&hist[(size_t)d * 288], d_hist, 288 * 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) { bail(FPNG_AMD_ERR_HIP); break; }
                }
            } while (false);
            if (two_pass && !meet([&] { // the image's histogram = the bands' sum (reference src/fpng.cpp:1021-1084 / :1299-1363 count the whole image)
                    for (uint32_t k = 0; k < nd; k++)
                        for (uint32_t i = 0; i < 288; i++) hist_sum[i] += hist[(size_t)k * 288 + i];
                }))
                return;
            std::memset(&stats[d], 0, sizeof stats[d]);
            if (!rcs[d] && rows) {
                int r2 = FPNG_AMD_OK;
                if (two_pass && hipMemcpyAsync(d_hist, hist_sum.data(), 288 * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess) r2 = FPNG_AMD_ERR_HIP;
                if (!r2) r2 = fpng_amd_band_encode(e, &band, flags & FPNG_AMD_ENCODE_SLOWER, two_pass ? d_hist : nullptr, &stats[d]); // (waits for the band's counts)
                if (r2) bail(r2);
            }
            // ---- the records meet: where every band starts, what the file's size and Adler-32 are, stored or not ----
            if (!meet([&] {
                    std::vector<fpng_amd_band_stats> st = stats;
                    for (uint32_t k = 0; k < nd; k++) // (bands without rows take their table facts from a band that has some)
                        if (!st[k].adler_len)
                            for (uint32_t q = 0; q < nd; q++)
                                if (stats[q].adler_len) st[k].first_token_bit = stats[q].first_token_bit, st[k].eob_bits = stats[q].eob_bits;
                    int r2 = fpng_amd_plan_bands(st.data(), nd, w, h, c, flags & FPNG_AMD_ENCODE_SLOWER, start_bits.data(), &plan);
                    if (!r2 && !plan.stored) {
                        out = reserve(user, kPngHeaderBytes + (size_t)plan.zlib_size + kPngTrailerBytes);
                        if (!out) r2 = fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "output buffer too small");
                    }
                    if (r2) rcs[0] = r2, errs[0] = fpng_amd_last_error(), abort_all = true;
                }))
                return;
            if (plan.stored || !rows) return;
            // ---- phase C: the band is placed at its bit position; its window goes down to its place in the file ----
            uint64_t fo = 0;
            size_t nbytes = 0;
            int r2 = fpng_amd_band_place(e, &band, start_bits[d], 0, e->d_stage_out.p, e->d_stage_out.cap, &fo, &nbytes);
            if (!r2) r2 = fpng_amd_band_crc(e, &raw[d], &ends[d]); // (waits for the placement)
            if (r2) { bail(r2); return; }
            const uint32_t head = (d && ((kPngHeaderBytes * 8 + start_bits[d]) & 127)) ? (uint32_t)std::min<size_t>(16, nbytes) : 0u; // the piece shared with the band in front
            seams[d].file_off = fo, seams[d].n = head;
            if ((head && hipMemcpyAsync(seams[d].b, e->d_stage_out.p, head, hipMemcpyDeviceToHost, e->stream) != hipSuccess) ||
                (nbytes > head && hipMemcpyAsync(out + fo + head, e->d_stage_out.p + head, nbytes - head, hipMemcpyDeviceToHost, e->stream) != hipSuccess) ||
                hipStreamSynchronize(e->stream) != hipSuccess)
                bail(FPNG_AMD_ERR_HIP);
        });
    for (auto &x : th) x.join();
    for (uint32_t d = 0; d < nd; d++)
        if (rcs[d]) return fail(rcs[d], errs[d].empty() ? "node image encode failed" : errs[d].c_str());
    if (plan.stored) // incompressible: the reference's stored-block outcome, decided and written by the whole-image path
        return fpng_amd_encode_host_to(node->enc[0], pixels, w, h, c, flags, reserve, user, out_size);
    for (uint32_t d = 0; d < nd; d++) // the pieces neighbouring windows share: each band wrote zeros where the other's bits are
        for (uint32_t i = 0; i < seams[d].n; i++) out[seams[d].file_off + i] |= seams[d].b[i];
    uint8_t head[58], tail[20];
    if ((rc = fpng_amd_png_head(w, h, c, plan.zlib_size, head))) return rc;
    fpng_amd_png_tail(plan.adler, fpng_amd_idat_crc_from_bands(raw.data(), ends.data(), nd, plan.zlib_size, plan.adler), tail);
    std::memcpy(out, head, kPngHeaderBytes);
    std::memcpy(out + kPngHeaderBytes + plan.zlib_size - 4, tail, 20);
    *out_size = kPngHeaderBytes + (size_t)plan.zlib_size + kPngTrailerBytes;
    return FPNG_AMD_OK;
}

int fpng_amd_node_encode_host_batch(fpng_amd_node *node, const fpng_amd_host_image *images, uint32_t n, uint32_t flags, int n_writers)
{
    if (!node || node->enc.empty() || !images || !n) return fail(FPNG_AMD_ERR_INVALID_ARG, "null/empty batch");
    const uint32_t nd = (uint32_t)node->enc.size();
    std::vector<std::vector<fpng_amd_host_image>> share(nd);
    for (uint32_t i = 0; i < n; i++) share[i % nd].push_back(images[i]); // (the records carry their own out_size pointers)
    std::vector<int> rcs(nd, FPNG_AMD_OK);
    std::vector<std::string> errs(nd);
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < nd; d++)
        if (!share[d].empty())
            th.emplace_back([&, d] {
                rcs[d] = fpng_amd_encode_host_batch(node->enc[d], share[d].data(), (uint32_t)share[d].size(), flags, n_writers);
                if (rcs[d]) errs[d] = fpng_amd_last_error(); // (the error text is per thread)
            });
    for (auto &t : th) t.join();
    for (uint32_t d = 0; d < nd; d++)
        if (rcs[d]) return fail(rcs[d], errs[d].c_str());
    return FPNG_AMD_OK;
}

} // extern "C"
