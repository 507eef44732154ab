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

// size() of a vector that held `prefill` bytes before the call, after fpng_decode_memory() returned *status (what a caller who
// reuses one vector observes on every exit: src/fpng.cpp:3087 empties it, :3111 sizes it)
size_t ref_decode_vector_size(const void *png, uint32_t size, uint32_t desired, size_t prefill, int *status)
{
    std::vector<uint8_t> v(prefill, 0xAB);
    uint32_t w, h, c;
    *status = fpng::fpng_decode_memory(png, size, v, w, h, c, desired);
    return v.size();
}

// Timed loop for the decoder's CPU baseline: `reps` decodes of the same file into one reused vector (as
// src/fpng_test.cpp:1236-1273 does), returns best seconds per decode (negative: the decoder's status code).
double ref_time_decode(const void *png, uint32_t size, uint32_t desired, int reps)
{
    std::vector<uint8_t> v;
    double best = 1e30;
    for (int i = 0; i < reps; i++) {
        uint32_t w, h, c;
        auto t0 = std::chrono::steady_clock::now();
        const int st = fpng::fpng_decode_memory(png, size, v, w, h, c, desired);
        auto t1 = std::chrono::steady_clock::now();
        if (st) return -(double)st;
        double s = std::chrono::duration<double>(t1 - t0).count();
        if (s < best) best = s;
    }
    return best;
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

#if FPNG_TRAIN_HUFFMAN_TABLES
// The reference's training mode (src/fpng_test.cpp:766-973) for images in memory: every image encoded with FPNG_ENCODE_SLOWER,
// fpng::g_huff_counts summed, then fpng::create_dynamic_block_prefix (src/fpng.cpp:910-988).  Built only into
// _ref/libfpng_ref_train.so (-DFPNG_TRAIN_HUFFMAN_TABLES=1).
int ref_train(const void *const *images, const uint32_t *w, const uint32_t *h, uint32_t n, uint32_t c, uint8_t *prefix, uint32_t prefix_cap,
              uint32_t *prefix_len, uint32_t *bit_buf, uint32_t *bit_buf_size, uint32_t codes[288], uint8_t code_sizes[288])
{
    uint64_t freq[fpng::HUFF_COUNTS_SIZE];
    memset(freq, 0, sizeof freq);
    for (uint32_t k = 0; k < n; k++) {
        memset(fpng::g_huff_counts, 0, sizeof(fpng::g_huff_counts));
        std::vector<uint8_t> v;
        if (!fpng::fpng_encode_image_to_memory(images[k], w[k], h[k], c, v, fpng::FPNG_ENCODE_SLOWER)) return 0;
        for (uint32_t i = 0; i < fpng::HUFF_COUNTS_SIZE; i++) freq[i] += fpng::g_huff_counts[i];
    }
    std::vector<uint8_t> p;
    uint64_t bb = 0;
    int bbs = 0;
    if (!fpng::create_dynamic_block_prefix(freq, c, p, bb, bbs, codes, code_sizes)) return 0;
    if (p.size() > prefix_cap) return 0;
    memcpy(prefix, p.data(), p.size());
    *prefix_len = (uint32_t)p.size(), *bit_buf = (uint32_t)bb, *bit_buf_size = (uint32_t)bbs;
    return 1;
}
#endif

} // extern "C"
