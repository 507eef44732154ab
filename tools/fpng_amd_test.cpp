// fpng_amd_test -- the fpng_test workflow (reference src/fpng_test.cpp:975-1639) pointed at the MI355X path.
//
//   fpng_amd_test [options] <input> [alpha source .png]
//     <input>   synth:<noise|solid|grad|blocks>:<W>x<H>x<C>[:seed]   (SURVEY.md B.1 generators)
//               or ANY non-interlaced .png (tools/png_loader.h: the role lodepng plays in the reference's harness,
//               fpng_test.cpp:1116-1122).  As there: the image is encoded with 4 channels if any alpha value is below 255,
//               else with 3; a second file name supplies the alpha channel from its GREEN channel (fpng_test.cpp:1124-1145)
//     -a        alpha = green (fpng_test.cpp:1147-1152)
//     -t        training mode: <input> = @filelist.txt (one .png per line); prints new 1-pass tables for the opaque and the
//               translucent files in the reference's format (fpng_test.cpp:766-973; there only in FPNG_TRAIN_HUFFMAN_TABLES builds)
//     -s        2-pass compression (FPNG_ENCODE_SLOWER)            reference fpng_test.cpp:1027
//     -u        stored Deflate blocks (FPNG_FORCE_UNCOMPRESSED)    reference fpng_test.cpp:1031
//     -c        one line of comma separated values                 reference fpng_test.cpp:1608-1633
//     -e        fuzz the encoder by mutating the input's pixels    reference fpng_test.cpp:381-615
//     -E        fuzz with random-noise images of random size       reference fpng_test.cpp:617-682
//     -f        decoder fuzz mode: <input> is an fpng-written .png that goes to fpng::fpng_decode_memory as it is (files of 256K
//               pixels and more: the GPU decoder); failure -> "fpng::fpng_decode() failed with error N!" and EXIT_FAILURE, else
//               out.png is written -- the mode the reference's README drives with zzuf (fpng_test.cpp:1092-1114).  With --judge
//               the reference's decoder (ref_decode) sees the same bytes: a different status or different pixels -> exit code 2
//     -n N      fuzz trials (default 1000, as the reference)       -m D  largest dimension for -E (default 8193)
//     -b N      also time N frames per call through fpng_amd_encode_host_batch (overlapped copies) and N
//               device-resident frames per submission (kernel-only rate)
//     -o file   write the encoded file (default fpng.png, as the reference)
//     --judge lib.so   a CPU encoder to compare every output byte for byte against: a library exporting
//               ref_encode() (the reference build) or fpo_encode() (the C restatement).  With -p N the same library
//               gives the CPU baseline: N threads, one image each (whole-node figure).
//
// Every encode goes through the drop-in `namespace fpng` (include/fpng.h -> libfpng.so -> C ABI -> HIP kernels);
// decode and verification as the reference does them: best of 3 encodes, best of 5 decodes, MP/s = 2^20 pixels per
// second like the reference prints (fpng_test.cpp:1212) with the 10^6 figure next to it.
#include "fpng.h"
#include "fpng_amd.h"
#include "png_loader.h"

#include <hip/hip_runtime_api.h>

#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef int (*judge_fn)(const void *, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t *, size_t, size_t *);

struct Options {
    bool slower = false, uncompressed = false, csv = false, fuzz = false, fuzz2 = false, green_to_alpha = false, train = false, fuzz_decoder = false;
    uint32_t trials = 1000, max_dim = 8193, batch = 0, cpu_threads = 0;
    const char *input = nullptr, *alpha_input = nullptr, *out = "fpng.png", *judge_path = nullptr;
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Rng { // xorshift32, as the synthetic generators
    uint32_t s;
    explicit Rng(uint32_t seed) : s(seed ? seed : 1u) {}
    uint32_t next()
    {
        s ^= s << 13, s ^= s >> 17, s ^= s << 5;
        return s;
    }
    uint32_t range(uint32_t lo, uint32_t hi) { return lo + next() % (hi - lo); } // [lo, hi)
    double unit() { return next() / 4294967296.0; }
};

bool load_input(const Options &o, std::vector<uint8_t> &px, uint32_t &w, uint32_t &h, uint32_t &c)
{
    const char *spec = o.input;
    if (!strncmp(spec, "synth:", 6)) {
        char kind[16] = {0};
        unsigned seed = 12345;
        if (sscanf(spec + 6, "%15[a-z]:%ux%ux%u:%u", kind, &w, &h, &c, &seed) < 4) return false;
        const int k = !strcmp(kind, "noise") ? 0 : !strcmp(kind, "solid") ? 1 : !strcmp(kind, "grad") ? 2 : !strcmp(kind, "blocks") ? 3 : -1;
        if (k < 0 || (c != 3 && c != 4)) return false;
        px.resize((size_t)w * h * c);
        return fpng_amd_synth_image(k, seed, w, h, c, px.data()) == 0;
    }
    // any PNG -> RGBA8 (reference fpng_test.cpp:1116-1122)
    std::vector<uint8_t> file, rgba;
    std::string err;
    if (!png_loader::read_file(spec, file) || !png_loader::load_rgba(file.data(), file.size(), rgba, w, h, err)) {
        fprintf(stderr, "Failed unpacking source file \"%s\"%s%s\n", spec, err.empty() ? "" : ": ", err.c_str());
        return false;
    }
    if (o.alpha_input) { // alpha from the green channel of a second image (fpng_test.cpp:1124-1145)
        std::vector<uint8_t> afile, a;
        uint32_t aw = 0, ah = 0;
        if (!png_loader::read_file(o.alpha_input, afile) || !png_loader::load_rgba(afile.data(), afile.size(), a, aw, ah, err)) {
            fprintf(stderr, "Failed unpacking alpha source file \"%s\"\n", o.alpha_input);
            return false;
        }
        for (uint32_t y = 0; y < std::min(ah, h); y++)
            for (uint32_t x = 0; x < std::min(aw, w); x++) rgba[((size_t)y * w + x) * 4 + 3] = a[((size_t)y * aw + x) * 4 + 1];
    } else if (o.green_to_alpha) { // fpng_test.cpp:1147-1152
        for (size_t i = 0; i < (size_t)w * h; i++) rgba[i * 4 + 3] = rgba[i * 4 + 1];
    }
    bool has_alpha = false; // fpng_test.cpp:1154-1166: 32 bpp only if some pixel is not opaque
    for (size_t i = 0; i < (size_t)w * h && !has_alpha; i++) has_alpha = rgba[i * 4 + 3] < 255;
    c = has_alpha ? 4 : 3;
    if (has_alpha)
        px.swap(rgba);
    else {
        px.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; i++) memcpy(&px[i * 3], &rgba[i * 4], 3);
    }
    return true;
}

bool encode_checked(const uint8_t *px, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, judge_fn judge, std::vector<uint8_t> &png,
                    const char *what)
{
    if (!fpng::fpng_encode_image_to_memory(px, w, h, c, png, flags)) {
        fprintf(stderr, "%s: fpng_encode_image_to_memory() failed!\n", what);
        return false;
    }
    std::vector<uint8_t> dec;
    uint32_t dw, dh, dc;
    if (fpng::fpng_decode_memory(png.data(), (uint32_t)png.size(), dec, dw, dh, dc, c) != fpng::FPNG_DECODE_SUCCESS || dw != w || dh != h ||
        dc != c || memcmp(dec.data(), px, (size_t)w * h * c) != 0) {
        fprintf(stderr, "%s: decoded image failed verification\n", what);
        return false;
    }
    if (judge) {
        std::vector<uint8_t> ref(fpng_amd_max_encoded_size(w, h, c) + 64);
        size_t n = 0;
        if (!judge(px, w, h, c, flags, ref.data(), ref.size(), &n) || n != png.size() || memcmp(ref.data(), png.data(), n) != 0) {
            fprintf(stderr, "%s: output differs from the CPU encoder's (%zu vs %zu bytes)\n", what, png.size(), n);
            return false;
        }
    }
    return true;
}

// reference fuzz_test_encoder (fpng_test.cpp:381-615): mutate the source image, encode, decode, compare
int fuzz_encoder(const std::vector<uint8_t> &src, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, const Options &o, judge_fn judge)
{
    std::vector<uint8_t> tmp, png;
    std::vector<std::pair<std::string, uint32_t>> kinds;
    for (uint32_t trial = 0; trial < o.trials; trial++) {
        Rng r(trial + 1);
        tmp = src;
        // the reference's six kinds of damage with its probabilities (fpng_test.cpp:403-515); runs that are NOT pixel-aligned
        // (byte fill / byte xor) are what stresses the 3- and 4-byte pixel compare of the run finder
        const double rand_fract = 0.000001 + r.unit() * 0.099999;
        const char *kind;
        if (r.unit() < 0.05) { // colour fill runs over the whole image, 1..32 pixels each
            kind = "color fill runs";
            for (size_t ofs = 0; ofs < tmp.size();) {
                const uint32_t left = (uint32_t)((tmp.size() - ofs) / c), run = r.range(1, std::min(left, 32u) + 1);
                uint8_t lit[4] = {(uint8_t)r.next(), (uint8_t)r.next(), (uint8_t)r.next(), (uint8_t)r.next()};
                for (uint32_t i = 0; i < run; i++, ofs += c) memcpy(&tmp[ofs], lit, c);
            }
        } else if (r.unit() < 0.05) { // colour runs, one in five xor-ed with a colour, the others left alone
            kind = "color corrupt runs";
            for (size_t ofs = 0; ofs < tmp.size();) {
                const uint32_t left = (uint32_t)((tmp.size() - ofs) / c), run = r.range(1, std::min(left, 32u) + 1);
                uint8_t lit[4] = {(uint8_t)r.next(), (uint8_t)r.next(), (uint8_t)r.next(), (uint8_t)r.next()};
                const bool hit = r.unit() > 0.8;
                for (uint32_t i = 0; i < run; i++, ofs += c)
                    for (uint32_t j = 0; j < c && hit; j++) tmp[ofs + j] ^= lit[j];
            }
        } else if (r.unit() < 0.05) { // BYTE fill runs of 1..258 bytes
            kind = "fill runs";
            for (size_t ofs = 0; ofs < tmp.size();) {
                const uint32_t left = (uint32_t)std::min<size_t>(tmp.size() - ofs, 258), run = r.range(1, left + 1);
                memset(&tmp[ofs], (int)(r.next() & 0xFF), run);
                ofs += run;
            }
        } else if (r.unit() < 0.15) { // BYTE runs of 1..32 bytes, nine in ten xor-ed with a value
            kind = "corrupt runs";
            for (size_t ofs = 0; ofs < tmp.size();) {
                const uint32_t left = (uint32_t)std::min<size_t>(tmp.size() - ofs, 32), run = r.range(1, left + 1);
                if (r.unit() > 0.1) {
                    const uint8_t v = (uint8_t)r.next();
                    for (uint32_t i = 0; i < run; i++) tmp[ofs + i] ^= v;
                }
                ofs += run;
            }
        } else if (r.unit() < 0.005) {
            kind = "full random";
            for (auto &b : tmp) b = (uint8_t)r.next();
        } else { // every bit flipped with probability rand_fract
            kind = "bits flipped";
            const uint32_t thresh = (uint32_t)(4294967295.0 * rand_fract);
            for (auto &b : tmp)
                for (int j = 0; j < 8; j++)
                    if (r.next() <= thresh) b ^= (uint8_t)(1 << j);
        }
        char what[96];
        snprintf(what, sizeof what, "fuzz trial %u (%s)", trial, kind);
        if (!encode_checked(tmp.data(), w, h, c, flags, judge, png, what)) return EXIT_FAILURE;
        if (trial % 50 == 0) printf("%u: %s, %zu bytes ok\n", trial, kind, png.size());
        bool seen = false;
        for (auto &k : kinds)
            if (k.first == kind) k.second++, seen = true;
        if (!seen) kinds.push_back({kind, 1u});
    }
    printf("kinds of damage:");
    for (auto &k : kinds) printf(" %s x %u;", k.first.c_str(), k.second);
    printf("\n");
    printf("fuzz_test_encoder: %u trials ok%s\n", o.trials, judge ? " (byte-identical to the CPU encoder)" : "");
    return EXIT_SUCCESS;
}

// reference fuzz_test_encoder2 (fpng_test.cpp:617-682): random-noise images of random dimensions
int fuzz_encoder2(uint32_t flags, const Options &o, judge_fn judge)
{
    Rng r(1);
    std::vector<uint8_t> px, png;
    for (uint32_t trial = 0; trial < o.trials; trial++) {
        const uint32_t w = r.range(1, o.max_dim + 1), h = r.range(1, o.max_dim + 1), c = (r.next() & 1) ? 4 : 3;
        px.resize((size_t)w * h * c);
        uint8_t *p = px.data();
        for (size_t i = 0; i < (size_t)w * h; i++) {
            const uint32_t v = r.next();
            *p++ = (uint8_t)v, *p++ = (uint8_t)(v >> 8), *p++ = (uint8_t)(v >> 16);
            if (c == 4) *p++ = (uint8_t)(v >> 24);
        }
        char what[64];
        snprintf(what, sizeof what, "Testing %ux%u %u", w, h, c);
        if (!encode_checked(px.data(), w, h, c, flags, judge, png, what)) return EXIT_FAILURE;
        printf("%s: fpng size %zu\n", what, png.size());
    }
    printf("fuzz_test_encoder2: %u trials ok%s\n", o.trials, judge ? " (byte-identical to the CPU encoder)" : "");
    return EXIT_SUCCESS;
}

// N frames per call: host batch (PCIe inclusive, overlapped) and device-resident submissions (kernel-only)
void batch_rates(const std::vector<uint8_t> &px, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, uint32_t n, double &host_s, double &dev_s)
{
    host_s = dev_s = -1;
    fpng_amd_encoder *enc = nullptr;
    if (fpng_amd_encoder_create(&enc, -1, nullptr)) return;
    const size_t cap = fpng_amd_max_encoded_size(w, h, c) + 64;
    std::vector<std::vector<uint8_t>> outs(n, std::vector<uint8_t>(cap));
    std::vector<size_t> sizes(n);
    std::vector<fpng_amd_host_image> hi(n);
    for (uint32_t i = 0; i < n; i++) {
        memset(&hi[i], 0, sizeof hi[i]);
        hi[i].pixels = px.data(), hi[i].w = w, hi[i].h = h, hi[i].num_chans = c;
        hi[i].out = outs[i].data(), hi[i].out_cap = cap, hi[i].out_size = &sizes[i];
    }
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now();
        if (fpng_amd_encode_host_batch(enc, hi.data(), n, flags, 0)) break;
        const double t = (now() - t0) / n;
        host_s = (host_s < 0 || t < host_s) ? t : host_s;
    }
    // device-resident: n copies of the frame, one submission per timing
    std::vector<void *> d_in(n), d_out(n);
    std::vector<fpng_amd_image> im(n);
    bool ok = true;
    for (uint32_t i = 0; i < n && ok; i++) {
        ok = hipMalloc(&d_in[i], px.size()) == hipSuccess && hipMalloc(&d_out[i], cap) == hipSuccess &&
             hipMemcpy(d_in[i], px.data(), px.size(), hipMemcpyHostToDevice) == hipSuccess;
        im[i].d_pixels = d_in[i], im[i].w = w, im[i].h = h, im[i].num_chans = c, im[i].d_out = (uint8_t *)d_out[i], im[i].out_cap = cap;
    }
    for (int rep = 0; rep < 12 && ok; rep++) {
        const double t0 = now();
        if (fpng_amd_encode_batch_async(enc, im.data(), n, flags) || fpng_amd_encode_finish(enc, nullptr, 0)) break;
        const double t = (now() - t0) / n;
        if (rep >= 2) dev_s = (dev_s < 0 || t < dev_s) ? t : dev_s;
    }
    for (uint32_t i = 0; i < n; i++) (void)hipFree(d_in[i]), (void)hipFree(d_out[i]);
    fpng_amd_encoder_destroy(enc);
}

// reference training_mode (fpng_test.cpp:766-973): the files of @list, split into opaque (24 bpp) and translucent (32 bpp) ones,
// each class through fpng_amd_train_tables (histograms, adjustment and table builder on the GPU); output in the reference's form
int training_mode(const Options &o)
{
    if (!o.input || o.input[0] != '@') {
        fprintf(stderr, "Must specify list of files to read using @filelist.txt\n");
        return EXIT_FAILURE;
    }
    FILE *lf = fopen(o.input + 1, "r");
    if (!lf) {
        fprintf(stderr, "Failed opening listing file %s\n", o.input + 1);
        return EXIT_FAILURE;
    }
    struct Dev { void *p; uint32_t w, h; };
    std::vector<Dev> cls[5];
    uint32_t failed = 0;
    char line[4096];
    while (fgets(line, sizeof line, lf)) {
        std::string name(line);
        while (!name.empty() && (name.back() == '\n' || name.back() == '\r' || name.back() == ' ')) name.pop_back();
        if (name.empty()) continue;
        printf("Processing file \"%s\"\n", name.c_str());
        Options one = o;
        one.input = name.c_str();
        std::vector<uint8_t> px;
        uint32_t w = 0, h = 0, c = 0;
        if (!load_input(one, px, w, h, c)) {
            fprintf(stderr, "WARNING: Failed unpacking source file \"%s\"! Skipping.\n", name.c_str());
            failed++;
            continue;
        }
        printf("Dimensions: %ux%u, Has Alpha: %u\n", w, h, c == 4);
        Dev d = {nullptr, w, h};
        if (hipMalloc(&d.p, px.size()) != hipSuccess || hipMemcpy(d.p, px.data(), px.size(), hipMemcpyHostToDevice) != hipSuccess) return EXIT_FAILURE;
        cls[c].push_back(d);
    }
    fclose(lf);
    printf("Total alpha files: %zu\nTotal opaque files: %zu\nTotal failed loading: %u\n", cls[4].size(), cls[3].size(), failed);
    if (cls[3].empty() && cls[4].empty()) return EXIT_FAILURE;
    fpng_amd_encoder *enc = nullptr;
    if (fpng_amd_encoder_create(&enc, -1, nullptr)) return EXIT_FAILURE;
    for (uint32_t c = 3; c <= 4; c++) {
        if (cls[c].empty()) continue;
        std::vector<fpng_amd_image> im(cls[c].size());
        for (size_t i = 0; i < im.size(); i++) im[i] = {cls[c][i].p, cls[c][i].w, cls[c][i].h, c, nullptr, 0};
        uint8_t prefix[512];
        size_t nb = 0;
        uint32_t bb = 0, bbs = 0, codes[288];
        uint8_t sizes[288];
        if (fpng_amd_train_tables(enc, im.data(), (uint32_t)im.size(), c, prefix, sizeof prefix, &nb, &bb, &bbs, codes, sizes)) {
            fprintf(stderr, "fpng_amd_train_tables() failed: %s\n", fpng_amd_last_error());
            return EXIT_FAILURE;
        }
        printf("\nstatic const uint8_t g_dyn_huff_%u[] = {\n", c);
        for (size_t i = 0; i < nb; i++) printf("%u%c %s", prefix[i], i + 1 != nb ? ',' : ' ', (i & 31) == 31 ? "\n" : "");
        printf("};\nconst uint32_t DYN_HUFF_%u_BITBUF = %u, DYN_HUFF_%u_BITBUF_SIZE = %u;\n", c, bb, c, bbs);
        printf("static const struct { uint8_t m_code_size; uint16_t m_code; } g_dyn_huff_%u_codes[288] = {\n", c);
        for (uint32_t i = 0; i < 288; i++) printf("{%u,%u}%c%s", sizes[i], codes[i], i != 287 ? ',' : ' ', (i & 31) == 31 ? "\n" : "");
        printf("};\n");
    }
    fpng_amd_encoder_destroy(enc);
    return EXIT_SUCCESS;
}

} // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strcmp(a, "--judge") && i + 1 < argc)
            o.judge_path = argv[++i];
        else if (a[0] == '-' && a[1] && !a[2]) {
            switch (a[1]) {
            case 's': o.slower = true; break;
            case 'u': o.uncompressed = true; break;
            case 'c': o.csv = true; break;
            case 'a': o.green_to_alpha = true; break;
            case 't': o.train = true; break;
            case 'e': o.fuzz = true; break;
            case 'E': o.fuzz2 = true; break;
            case 'f': o.fuzz_decoder = true; break;
            case 'n': o.trials = (uint32_t)atoi(argv[++i]); break;
            case 'm': o.max_dim = (uint32_t)atoi(argv[++i]); break;
            case 'b': o.batch = (uint32_t)atoi(argv[++i]); break;
            case 'p': o.cpu_threads = (uint32_t)atoi(argv[++i]); break;
            case 'o': o.out = argv[++i]; break;
            default: fprintf(stderr, "Unrecognized option: %s\n", a); return EXIT_FAILURE;
            }
        } else if (!o.input)
            o.input = a;
        else
            o.alpha_input = a;
    }
    if (!o.input && !o.fuzz2) {
        printf("Usage: fpng_amd_test [-s] [-u] [-a] [-c] [-e] [-E] [-f] [-t @filelist.txt] [-n trials] [-m maxdim] [-b frames] [-p cpu threads] [-o out.png] "
               "[--judge cpu_encoder.so] <synth:kind:WxHxC[:seed] | file.png> [alpha_file.png]\n");
        return EXIT_FAILURE;
    }
    fpng::fpng_init();
    if (!fpng::fpng_cpu_supports_sse41()) { // (the drop-in answers "is the MI355X path usable")
        fprintf(stderr, "no usable GPU: fpng_amd has no CPU path\n");
        return EXIT_FAILURE;
    }
    judge_fn judge = nullptr;
    typedef int (*judge_decode_fn)(const void *, uint32_t, uint8_t *, size_t, uint32_t *, uint32_t *, uint32_t *, uint32_t);
    judge_decode_fn judge_decode = nullptr;
    if (o.judge_path) {
        void *lib = dlopen(o.judge_path, RTLD_NOW);
        if (lib) {
            if (void (*init)() = (void (*)())dlsym(lib, "ref_init")) init();
            judge = (judge_fn)dlsym(lib, "ref_encode");
            if (!judge) judge = (judge_fn)dlsym(lib, "fpo_encode");
            judge_decode = (judge_decode_fn)dlsym(lib, "ref_decode");
        }
        if (!judge) {
            fprintf(stderr, "cannot use %s as judge: %s\n", o.judge_path, dlerror());
            return EXIT_FAILURE;
        }
    }
    const uint32_t flags = (o.slower ? fpng::FPNG_ENCODE_SLOWER : 0) | (o.uncompressed ? fpng::FPNG_FORCE_UNCOMPRESSED : 0);
    if (o.train) return training_mode(o);
    if (o.fuzz2) return fuzz_encoder2(flags, o, judge);
    if (o.fuzz_decoder) { // reference fpng_test.cpp:1092-1114
        FILE *f = fopen(o.input, "rb");
        if (!f) {
            fprintf(stderr, "Failed reading source file data \"%s\"\n", o.input);
            return EXIT_FAILURE;
        }
        std::vector<uint8_t> file;
        uint8_t chunk[65536];
        for (size_t n; (n = fread(chunk, 1, sizeof chunk, f)) > 0;) file.insert(file.end(), chunk, chunk + n);
        fclose(f);
        std::vector<uint8_t> pixels;
        uint32_t dw = 0, dh = 0, dc = 0;
        const int res = fpng::fpng_decode_memory(file.data(), (uint32_t)file.size(), pixels, dw, dh, dc, 3);
        if (judge_decode) { // the same bytes through the reference's decoder: status and pixels must agree
            uint32_t rw = 0, rh = 0, rc = 0;
            const size_t cap = res == 0 ? pixels.size() : ((size_t)1 << 28);
            std::vector<uint8_t> want(cap);
            const int rres = judge_decode(file.data(), (uint32_t)file.size(), want.data(), want.size(), &rw, &rh, &rc, 3);
            if (rres != res || (res == 0 && (rw != dw || rh != dh || rc != dc || memcmp(want.data(), pixels.data(), pixels.size()) != 0))) {
                fprintf(stderr, "DECODER MISMATCH: status %i vs the reference's %i (or different pixels)\n", res, rres);
                return 2;
            }
        }
        if (res != 0) {
            fprintf(stderr, "fpng::fpng_decode() failed with error %i!\n", res);
            return EXIT_FAILURE;
        }
        if (!fpng::fpng_encode_image_to_file("out.png", pixels.data(), dw, dh, 3, 0)) {
            fprintf(stderr, "writing out.png failed\n");
            return EXIT_FAILURE;
        }
        printf("Wrote out.png %ux%u %u\n", dw, dh, dc);
        return EXIT_SUCCESS;
    }

    std::vector<uint8_t> px;
    uint32_t w = 0, h = 0, c = 0;
    if (!load_input(o, px, w, h, c)) {
        fprintf(stderr, "Failed loading %s\n", o.input);
        return EXIT_FAILURE;
    }
    if (!o.csv) printf("Dimensions: %ux%u, %u channels, total pixels: %llu\n", w, h, c, (unsigned long long)w * h);
    if (o.fuzz) return fuzz_encoder(px, w, h, c, flags, o, judge);

    // ---- encode: best of 3, one reused vector (reference fpng_test.cpp:1181-1209) ----
    std::vector<uint8_t> png;
    double enc_s = 1e9;
    for (int i = 0; i < 3; i++) {
        const double t0 = now();
        if (!fpng::fpng_encode_image_to_memory(px.data(), w, h, c, png, flags)) {
            fprintf(stderr, "fpng_encode_image_to_memory() failed!\n");
            return EXIT_FAILURE;
        }
        enc_s = std::min(enc_s, now() - t0);
    }
    if (!encode_checked(px.data(), w, h, c, flags, judge, png, "verification")) return EXIT_FAILURE;
    if (FILE *f = fopen(o.out, "wb")) {
        fwrite(png.data(), 1, png.size(), f);
        fclose(f);
    }
    // ---- decode: best of 5 (reference fpng_test.cpp:1182, :1236-1262) ----
    std::vector<uint8_t> dec;
    double dec_s = 1e9;
    for (int i = 0; i < 5; i++) {
        uint32_t dw, dh, dc;
        const double t0 = now();
        if (fpng::fpng_decode_memory(png.data(), (uint32_t)png.size(), dec, dw, dh, dc, c) != fpng::FPNG_DECODE_SUCCESS) return EXIT_FAILURE;
        dec_s = std::min(dec_s, now() - t0);
    }
    double host_batch_s = -1, dev_s = -1;
    if (o.batch) batch_rates(px, w, h, c, flags, o.batch, host_batch_s, dev_s);
    // ---- CPU baseline through the judge library: N threads, one image each ----
    double cpu1_s = -1, cpuN_s = -1;
    if (judge && o.cpu_threads) {
        auto run = [&](uint32_t threads) {
            std::vector<std::thread> ts;
            std::vector<double> best(threads, 1e9);
            for (uint32_t t = 0; t < threads; t++)
                ts.emplace_back([&, t] {
                    std::vector<uint8_t> out(fpng_amd_max_encoded_size(w, h, c) + 64);
                    size_t n = 0;
                    for (int rep = 0; rep < 3; rep++) {
                        const double t0 = now();
                        judge(px.data(), w, h, c, flags, out.data(), out.size(), &n);
                        best[t] = std::min(best[t], now() - t0);
                    }
                });
            for (auto &t : ts) t.join();
            double worst = 0;
            for (double b : best) worst = std::max(worst, b);
            return worst;
        };
        cpu1_s = run(1);
        cpuN_s = run(o.cpu_threads) / o.cpu_threads; // seconds per image with all threads busy
    }
    const double mip = (double)w * h / (1024.0 * 1024.0), mp = (double)w * h / 1e6;
    if (o.csv) {
        // file, w, h, chans, fpng: encode s, size MiB, decode s, encode MiP/s, decode MiP/s (the reference's fpng columns), then
        // encode MP/s (10^6), host-batch s/frame, device-resident s/frame, CPU 1-thread s, CPU N-thread s/image
        printf("%s, %u, %u, %u,    %f, %f, %f, %4.3f, %4.3f,    %4.3f, %f, %f, %f, %f\n", o.input, w, h, c, enc_s, png.size() / (1024.0 * 1024.0),
               dec_s, mip / enc_s, mip / dec_s, mp / enc_s, host_batch_s, dev_s, cpu1_s, cpuN_s);
        return EXIT_SUCCESS;
    }
    printf("** Encoding:\nFPNG (MI355X, drop-in, PCIe inclusive): %4.6f secs, %zu bytes, %4.3f MB, %4.3f MiP/sec (%4.3f MP/sec)\n", enc_s, png.size(),
           png.size() / (1024.0 * 1024.0), mip / enc_s, mp / enc_s);
    if (host_batch_s > 0) printf("  %u frames per call, copies overlapped:   %4.6f secs/frame, %4.3f MP/sec\n", o.batch, host_batch_s, mp / host_batch_s);
    if (dev_s > 0) printf("  %u device-resident frames per submission: %4.6f secs/frame, %4.3f MP/sec\n", o.batch, dev_s, mp / dev_s);
    if (cpu1_s > 0)
        printf("CPU encoder (%s): 1 thread %4.6f secs (%4.3f MP/sec); %u threads, one image each: %4.3f MP/sec in total\n", o.judge_path, cpu1_s,
               mp / cpu1_s, o.cpu_threads, mp / cpuN_s);
    printf("** Decoding:\nFPNG (CPU): %3.6f secs, %4.3f MiP/sec\n", dec_s, mip / dec_s);
    printf("Wrote %s; decode verified%s\n", o.out, judge ? "; bytes identical to the CPU encoder's" : "");
    return EXIT_SUCCESS;
}
