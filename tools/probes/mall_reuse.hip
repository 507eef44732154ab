// mall_reuse.hip -- round 5 probe: does the 256 MiB Infinity Cache of one MI355X serve a producer -> consumer pair of kernels?
// The encode chain writes the rows' local streams (encode_rows) and reads them back (assemble); the decoder writes the filtered bytes
// (dec_emit) and reads them back (dec_unfilter): if a buffer of S bytes written by one kernel is read by the next one from the cache,
// the order of the work (image by image instead of batch by batch) decides the HBM traffic.
//   1. read S bytes again and again (S = 16 MiB .. 1 GiB): the L2 / Infinity Cache / HBM plateaus of a read
//   2. write S bytes, then read them: the read's rate
//   3. write S bytes, stream P bytes of OTHER data through (read, default or nontemporal), then read the S bytes
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mall_reuse tools/probes/mall_reuse.hip ; ./mall_reuse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool NT> __global__ __launch_bounds__(256) void rd(const u32x4 *src, size_t n, uint32_t *sink)
{
    u32x4 acc = {};
    const size_t base = (size_t)blockIdx.x * 1024;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < n) acc ^= NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
    if (acc.x == 0x12345u) sink[0] = 1;
}
template <bool NT> __global__ __launch_bounds__(256) void wr(u32x4 *dst, size_t n, u32x4 v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}

int main()
{
    const size_t big = 2ull << 30;
    uint8_t *a, *b; uint32_t *sink;
    CHECK(hipMalloc(&a, big)); CHECK(hipMalloc(&b, big)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(a, 1, big)); CHECK(hipMemset(b, 2, big));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const u32x4 v4 = {1, 2, 3, 4};
    auto launch_rd = [&](const uint8_t *p, size_t bytes, bool nt) {
        const size_t n = bytes / 16; const unsigned g = (unsigned)((n + 1023) / 1024);
        if (nt) rd<true><<<g, 256>>>((const u32x4 *)p, n, sink); else rd<false><<<g, 256>>>((const u32x4 *)p, n, sink);
    };
    auto launch_wr = [&](uint8_t *p, size_t bytes, bool nt) {
        const size_t n = bytes / 16; const unsigned g = (unsigned)((n + 255) / 256);
        if (nt) wr<true><<<g, 256>>>((u32x4 *)p, n, v4); else wr<false><<<g, 256>>>((u32x4 *)p, n, v4);
    };
    // best-of-5 of the LAST kernel of `pre(); timed();`
    auto time_last = [&](auto pre, auto timed) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            pre(); hipEventRecord(e0, 0); timed(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        return best;
    };
    printf("1. read S bytes that the SAME read kernel touched last (TB/s)\n");
    for (size_t mb = 16; mb <= 1024; mb *= 2) {
        const size_t s = mb << 20;
        const float t = time_last([&] { launch_rd(a, s, false); }, [&] { launch_rd(a, s, false); });
        const float tn = time_last([&] { launch_rd(a, s, true); }, [&] { launch_rd(a, s, true); });
        printf("  S = %5zu MiB: default %6.2f   nontemporal %6.2f\n", mb, s / t / 1e9, s / tn / 1e9); fflush(stdout);
    }
    printf("2. write S bytes, then read them (the READ's TB/s; columns: store default / nontemporal)\n");
    for (size_t mb = 16; mb <= 1024; mb *= 2) {
        const size_t s = mb << 20;
        const float t = time_last([&] { launch_wr(a, s, false); }, [&] { launch_rd(a, s, false); });
        const float tn = time_last([&] { launch_wr(a, s, true); }, [&] { launch_rd(a, s, false); });
        printf("  S = %5zu MiB: %6.2f   %6.2f\n", mb, s / t / 1e9, s / tn / 1e9); fflush(stdout);
    }
    printf("3. write S bytes, read P bytes of other data (default / nontemporal loads), then read the S bytes (TB/s of that read)\n");
    for (size_t mb = 32; mb <= 128; mb *= 2)
        for (size_t pmb = 64; pmb <= 1024; pmb *= 2) {
            const size_t s = mb << 20, p = pmb << 20;
            const float t = time_last([&] { launch_wr(a, s, false); launch_rd(b, p, false); }, [&] { launch_rd(a, s, false); });
            const float tn = time_last([&] { launch_wr(a, s, false); launch_rd(b, p, true); }, [&] { launch_rd(a, s, false); });
            printf("  S = %4zu MiB, P = %5zu MiB: %6.2f   %6.2f\n", mb, pmb, s / t / 1e9, s / tn / 1e9); fflush(stdout);
        }
    printf("4. the encode shape: read 2S (image) + write S (streams), then read S + write S elsewhere: time of the second pair, ms (HBM-only estimate: 2S / 5.5 TB/s)\n");
    for (size_t mb = 16; mb <= 512; mb *= 2) {
        const size_t s = mb << 20;
        const float t = time_last([&] { launch_rd(b, 2 * s, false); launch_wr(a, s, false); }, [&] { launch_rd(a, s, false); launch_wr(a + big / 2, s, false); });
        printf("  S = %5zu MiB: %7.3f ms   (%6.2f TB/s over 2S)\n", mb, t, 2.0 * s / t / 1e9); fflush(stdout);
    }
    return 0;
}
