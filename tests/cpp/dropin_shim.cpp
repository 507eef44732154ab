// Test-only C shim so Python can call the `namespace fpng` drop-in (std::vector API) via ctypes.
#include "fpng.h"
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
extern "C" {
// Timed loop through the drop-in exactly as the reference's harness times its encoder (fpng_test.cpp:1198-1209): `reps` calls
// of fpng::fpng_encode_image_to_memory into ONE reused vector (reuse = 1) or into a fresh vector per call (reuse = 0, the
// vector's construction and growth inside the timed region); best seconds per call.
double shim_time_encode(const void *img, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, int reps, int reuse, size_t *size)
{
    static std::vector<uint8_t> keep; // (lives across calls like a capture loop's buffer)
    double best = 1e30;
    for (int i = 0; i < reps; i++) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<uint8_t> fresh;
        std::vector<uint8_t> &v = reuse ? keep : fresh;
        if (!fpng::fpng_encode_image_to_memory(img, w, h, c, v, flags)) return -1.0;
        auto t1 = std::chrono::steady_clock::now();
        double s = std::chrono::duration<double>(t1 - t0).count();
        if (s < best) best = s;
        if (size) *size = v.size();
    }
    return best;
}
int shim_encode(const void *img, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, uint8_t *out, size_t cap, size_t *size)
{
    std::vector<uint8_t> v;
    if (!fpng::fpng_encode_image_to_memory(img, w, h, c, v, flags)) return 0;
    *size = v.size();
    if (v.size() > cap) return 0;
    memcpy(out, v.data(), v.size());
    return 1;
}
int shim_encode_file(const char *name, const void *img, uint32_t w, uint32_t h, uint32_t c, uint32_t flags)
{
    return fpng::fpng_encode_image_to_file(name, img, w, h, c, flags) ? 1 : 0;
}
int shim_get_info(const void *png, uint32_t size, uint32_t *w, uint32_t *h, uint32_t *c) { return fpng::fpng_get_info(png, size, *w, *h, *c); }
int shim_decode(const void *png, uint32_t size, uint8_t *out, size_t cap, uint32_t *w, uint32_t *h, uint32_t *c, uint32_t desired)
{
    std::vector<uint8_t> v;
    int st = fpng::fpng_decode_memory(png, size, v, *w, *h, *c, desired);
    if (st == 0) {
        if (v.size() > cap) return -1;
        memcpy(out, v.data(), v.size());
    }
    return st;
}
// size() of a vector that held `prefill` bytes before the call, after fpng_decode_memory() returned *status
size_t shim_decode_vector_size(const void *png, uint32_t size, uint32_t desired, size_t prefill, int *status)
{
    std::vector<uint8_t> v(prefill, 0xAB);
    uint32_t w, h, c;
    *status = fpng::fpng_decode_memory(png, size, v, w, h, c, desired);
    return v.size();
}
// best seconds per fpng::fpng_decode_memory() call into one reused std::vector (like shim_time_encode)
double shim_time_decode(const void *png, uint32_t size, uint32_t desired, int reps)
{
    static std::vector<uint8_t> keep;
    double best = 1e30;
    for (int i = 0; i < reps; i++) {
        uint32_t w, h, c;
        auto t0 = std::chrono::steady_clock::now();
        if (fpng::fpng_decode_memory(png, size, keep, w, h, c, desired) != 0) return -1.0;
        auto t1 = std::chrono::steady_clock::now();
        double s = std::chrono::duration<double>(t1 - t0).count();
        if (s < best) best = s;
    }
    return best;
}
extern "C" unsigned long long fpng_amd_dropin_gpu_decodes();
unsigned long long shim_gpu_decodes() { return fpng_amd_dropin_gpu_decodes(); }
int shim_decode_file(const char *name, uint8_t *out, size_t cap, uint32_t *w, uint32_t *h, uint32_t *c, uint32_t desired)
{
    std::vector<uint8_t> v;
    int st = fpng::fpng_decode_file(name, v, *w, *h, *c, desired);
    if (st == 0 && v.size() <= cap) memcpy(out, v.data(), v.size());
    return st;
}
// The reference's functions are re-entrant and callers encode from many threads at once (src/fpng.cpp has no locks): n_threads
// threads, each encoding ITS image `reps` times into its own vector; outs[t] (cap bytes each) / sizes[t] receive thread t's last
// file, `agree` = 1 when every repetition of every thread produced the same bytes as its first.
int shim_encode_threads(int n_threads, const void *const *imgs, const uint32_t *w, const uint32_t *h, const uint32_t *c, const uint32_t *flags,
                        int reps, uint8_t *const *outs, size_t cap, size_t *sizes, int *agree)
{
    std::vector<int> ok(n_threads, 1), same(n_threads, 1);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t] {
            std::vector<uint8_t> first, v;
            for (int r = 0; r < reps; r++) {
                if (!fpng::fpng_encode_image_to_memory(imgs[t], w[t], h[t], c[t], v, flags[t])) {
                    ok[t] = 0;
                    return;
                }
                if (r == 0)
                    first = v;
                else if (v != first)
                    same[t] = 0;
            }
            sizes[t] = v.size();
            if (v.size() <= cap) memcpy(outs[t], v.data(), v.size());
        });
    for (auto &x : th) x.join();
    *agree = 1;
    for (int t = 0; t < n_threads; t++) {
        if (!ok[t]) return 0;
        if (!same[t]) *agree = 0;
    }
    return 1;
}
// threads, each decoding ITS file `reps` times (fpng::fpng_decode_memory into its own vector); outs[t] (cap bytes each) receives
// thread t's last pixels, status[t] its last status code; `agree` = 1 when every repetition of a thread gave the same result.
int shim_decode_threads(int n_threads, const void *const *pngs, const uint32_t *sizes, uint32_t desired, int reps, uint8_t *const *outs, size_t cap, int *status,
                        size_t *out_sizes, int *agree)
{
    std::vector<int> same(n_threads, 1);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t] {
            std::vector<uint8_t> first, v;
            int st0 = 0;
            for (int r = 0; r < reps; r++) {
                uint32_t w, h, c;
                const int st = fpng::fpng_decode_memory(pngs[t], sizes[t], v, w, h, c, desired);
                if (r == 0)
                    first = v, st0 = st;
                else if (st != st0 || v != first)
                    same[t] = 0;
                status[t] = st;
            }
            out_sizes[t] = v.size();
            if (v.size() <= cap) memcpy(outs[t], v.data(), v.size());
        });
    for (auto &x : th) x.join();
    *agree = 1;
    for (int t = 0; t < n_threads; t++)
        if (!same[t]) *agree = 0;
    return 1;
}
void shim_init() { fpng::fpng_init(); }
int shim_supported() { return fpng::fpng_cpu_supports_sse41() ? 1 : 0; }
uint32_t shim_crc32(const void *p, size_t n, uint32_t prev) { return fpng::fpng_crc32(p, n, prev); }
uint32_t shim_adler32(const void *p, size_t n, uint32_t prev) { return fpng::fpng_adler32(p, n, prev); }
}
