// ref_shim.cpp -- C-ABI doorway into the UNMODIFIED reference (TEST INFRASTRUCTURE ONLY).
//
// Compiled by oracle/Makefile together with /root/reference/src/fpng.cpp (from where it lies,
// never copied) into oracle/_ref/libfpng_ref.so.  Used to (a) pin oracle/fpng_oracle.c, (b) make
// tests/golden/*, (c) serve as bench.py's cpu_baseline of kind "reference".
//
// The reference is built with -DNDEBUG so that its bad-argument paths return false instead of
// hitting assert(0) (src/fpng.cpp:1664-1680).
#include "fpng.h" // the reference's own header, found via -I/root/reference/src

#include <chrono>
#include <cstring>
#include <vector>

extern "C" {

void ref_init() { fpng::fpng_init(); }
int ref_supports_sse41() { return fpng::fpng_cpu_supports_sse41() ? 1 : 0; }
uint32_t ref_crc32(const void *p, size_t n, uint32_t prev) { return fpng::fpng_crc32(p, n, prev); }
uint32_t ref_adler32(const void *p, size_t n, uint32_t prev) { return fpng::fpng_adler32(p, n, prev); }

// returns 1/0 like the reference's bool; *out_size is the size needed/produced
int ref_encode(const void *img, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, uint8_t *out, size_t out_cap,
               size_t *out_size)
{
    std::vector<uint8_t> v;
    if (!fpng::fpng_encode_image_to_memory(img, w, h, c, v, flags)) return 0;
    *out_size = v.size();
    if (v.size() > out_cap) return 0;
    memcpy(out, v.data(), v.size());
    return 1;
}

int ref_get_info(const void *png, uint32_t size, uint32_t *w, uint32_t *h, uint32_t *c)
{
    return fpng::fpng_get_info(png, size, *w, *h, *c);
}

int ref_decode(const void *png, uint32_t size, uint8_t *out, size_t out_cap, uint32_t *w, uint32_t *h, uint32_t *c,
               uint32_t desired)
{
    std::vector<uint8_t> v;
    int st = fpng::fpng_decode_memory(png, size, v, *w, *h, *c, desired);
    if (st == 0) {
        if (v.size() > out_cap) return -1;
        memcpy(out, v.data(), v.size());
    }
    return st;
}

// Timed loop for the CPU baseline: `reps` encodes of the same image into one reused vector
// (as src/fpng_test.cpp:1198-1209 does), returns best seconds per encode.
double ref_time_encode(const void *img, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, int reps, size_t *out_size)
{
    std::vector<uint8_t> v;
    double best = 1e30;
    for (int i = 0; i < reps; i++) {
        auto t0 = std::chrono::steady_clock::now();
        if (!fpng::fpng_encode_image_to_memory(img, w, h, c, v, flags)) return -1.0;
        auto t1 = std::chrono::steady_clock::now();
        double s = std::chrono::duration<double>(t1 - t0).count();
        if (s < best) best = s;
    }
    if (out_size) *out_size = v.size();
    return best;
}

} // extern "C"
