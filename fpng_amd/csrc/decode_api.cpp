// decode_api.cpp -- host side of the GPU batch decoder (decode.hip): what the reference's fpng_decode_memory does in front of
// its pixel loops (reference src/fpng.cpp:2904-3222: container walk, IDAT checks, block type, dynamic header), then uploads (or,
// for files that already live in device memory, fetches the few hundred bytes it has to read) and launches.  The parsing code
// is the CPU decoder's own (png_parse.h), so the status codes of damaged containers are the same.
#include "decode.h"
#include "decode_core.h"
#include "encoder.h"
#include "host_workers.h"
#include "png_parse.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace fpng_amd;

namespace {

constexpr uint32_t kMaxGroups = 4;    // groups of files whose upload and decode overlap (8 measured: 5-15 % slower, a dozen launches per group)
constexpr uint32_t kBorderRounds = 2; // synchronisation rounds across workgroup borders launched without asking whether they are needed
constexpr uint32_t kMaxRounds = 64;   // ... and the most a file gets before it is left to the CPU decoder
constexpr uint32_t kHeadBytes = 1024, kTailBytes = 64; // what is copied back of a device-resident file before anything else

struct Parsed {
    uint32_t w = 0, h = 0, c = 0, idat_ofs = 0, idat_len = 0;
    int status = 0;       // fpng::FPNG_DECODE_*
    uint32_t mode = 0;    // 0 dynamic, 1 stored
    uint64_t first_bit = 0;
    int lut = -1;         // index into the unique lookup tables
};

// The stored-block layout the reference accepts (src/fpng.cpp:2107-2207): block headers, sizes, filter bytes 0, exact end.
// 0 = the encoder's own layout (full 65535-byte blocks, then the rest: what dec_stored_kernel copies), 1 = not acceptable,
// 2 = acceptable to the reference but cut into other block sizes (or with the one zero byte behind the image that the reference
// lets pass): left to the CPU decoder.
int check_stored(const uint8_t *z, uint32_t avail, uint32_t zlib_len, uint32_t w, uint32_t h, uint32_t c)
{
    const uint64_t stride = (uint64_t)w * c + 1, total = stride * h;
    uint64_t src = 2, got = 0;
    bool usual = true;
    for (;;) {
        if (src + 5 > avail) return 1;
        const bool final_block = z[src] & 1;
        if (((z[src] >> 1) & 3) != 0) return 1;
        const uint32_t len = z[src + 1] | (z[src + 2] << 8), nlen = z[src + 3] | (z[src + 4] << 8);
        src += 5;
        if (len != (~nlen & 0xFFFF) || src + len > avail) return 1;
        if (!final_block && len != 65535) usual = false;
        for (uint64_t r = (got + stride - 1) / stride * stride; r < got + len; r += stride) // filter bytes inside this block
            if (z[src + (r - got)] != 0) return 1;
        got += len;
        src += len;
        if (final_block) break;
    }
    if (src + 4 != zlib_len) return 1;
    // (the reference reads ONE byte more than the image holds if it is 0: it takes it for the filter byte of a row that never
    //  comes -- :2158-2166 look at a row's first byte before they ask whether there is room; the loop above has checked it)
    if (got == total + 1) return 2;
    if (got != total) return 1;
    return usual ? 0 : 2;
}

// The kernels' lookup table (decode_core.h) from the host parser's (symbol | code length << 9 per 12-bit index) and the code lengths:
// up to three literals per entry, length symbols 257..285 with their base length and extra bit count (RFC 1951 3.2.5; 286 / 287
// never occur in a valid stream), then the literals' code lengths.
void build_multi_lut(const uint32_t *table, const uint8_t sizes[288], uint32_t *lut)
{
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint32_t bits = fpng::parse::kTableBits;
    for (uint32_t k = 0; k < (1u << bits); k++) {
        const uint32_t e1 = table[k], l1 = (e1 >> 9) & 15u, s1 = e1 & 511u;
        uint32_t ent = 0;
        if (!l1 || s1 > 285)
            ent = 0;
        else if (s1 == 256)
            ent = dec::kEntEob | l1 << 12;
        else if (s1 > 256)
            ent = len_extra[s1 - 257] ? dec::kEntMatch | l1 << 12 | (uint32_t)len_extra[s1 - 257] << 9 | len_base[s1 - 257] : (l1 + 1) << 28 | dec::kEntMatch | len_base[s1 - 257];
        else {
            uint32_t L = l1, n = 1, lits = s1;
            while (n < 3) { // the next code is whole if its length fits into the index bits that are left
                const uint32_t e2 = table[k >> L], l2 = (e2 >> 9) & 15u, s2 = e2 & 511u;
                if (!l2 || s2 >= 256 || L + l2 > bits) break;
                lits |= s2 << (8 * n);
                n++, L += l2;
            }
            ent = L << 28 | n << 26 | lits;
        }
        lut[k] = ent;
    }
    uint8_t *lenof = (uint8_t *)(lut + dec::kLutEntries);
    std::memcpy(lenof, sizes, 256);
}

// What the host settles about one file's zlib stream before the GPU sees it (the container was parsed: p.w .. p.idat_len): stored
// blocks (checked here) or one final dynamic block (header read: `table` = the host parser's lookup table, `sizes` = the literal /
// length code lengths, p.first_bit = the first row token).  z / avail: the stream's bytes in host memory, `complete`: all of them
// (else only a head of the file: fpng::parse::kParseNeedMore asks for the rest).  Returns the reference's status code (0 or
// FPNG_DECODE_NOT_FPNG) or FPNG_AMD_DECODE_UNDECIDED (a stored layout only the CPU decoder takes).
// (memo: the last dynamic header read -- the files of a 1-pass batch all begin with the same one, and reading it costs 4 us a
//  file; a header that is bit for bit the memo's is not read again.  `table` may be nullptr: the code is then only checked)
struct HeaderMemo {
    uint32_t bits = 0, chans = 0; // bits: the header's end = the first row token, from the stream's first byte (0: no memo)
    uint8_t bytes[320];
    uint8_t sizes[288];
};
int plan_stream(const uint8_t *z, uint32_t avail, bool complete, Parsed &p, uint32_t *table, uint8_t sizes[288], HeaderMemo *memo = nullptr)
{
    using namespace fpng::parse;
    if (avail < 3) return complete ? (int)fpng::FPNG_DECODE_NOT_FPNG : kParseNeedMore;
    if (p.idat_len < 7 || z[0] != 0x78 || z[1] != 0x01) return fpng::FPNG_DECODE_NOT_FPNG;
    const uint64_t total = ((uint64_t)p.w * p.c + 1) * p.h;
    if ((z[2] & 6) == 0) {
        if (total > p.idat_len) return fpng::FPNG_DECODE_NOT_FPNG; // (stored blocks cannot hold the image)
        if (!complete) {
            // only a head of the file is here (it lies in device memory): with the size the USUAL layout has -- blocks of 65535
            // bytes and a last one -- dec_stored_kernel checks the block headers and the filter bytes where they are and reports
            // anything else (kDecStoredOdd: the file is then looked at on the host after all)
            const uint64_t nblk = (total + 65534) / 65535;
            if (p.idat_len != 2 + 5 * nblk + total + 4) return kParseNeedMore;
            p.mode = 1;
            return 0;
        }
        const int r = check_stored(z, avail, p.idat_len, p.w, p.h, p.c);
        if (r == 1) return fpng::FPNG_DECODE_NOT_FPNG;
        p.mode = 1;
        return r ? FPNG_AMD_DECODE_UNDECIDED : 0;
    }
    Bits in = {z, avail, 2, 0, 0, false};
    if (in.get(1) != 1 || in.get(2) != 2) return fpng::FPNG_DECODE_NOT_FPNG; // one final dynamic block
    bool known = false;
    if (memo && memo->bits && memo->chans == p.c) {
        const uint32_t nb = memo->bits >> 3, rb = memo->bits & 7;
        known = avail >= nb + 9 && !std::memcmp(z, memo->bytes, nb) && !((z[nb] ^ memo->bytes[nb]) & ((1u << rb) - 1u));
    }
    if (known) {
        std::memcpy(sizes, memo->sizes, 288);
        p.first_bit = memo->bits;
    } else {
        if (memo) memo->bits = 0;
        const bool ok = read_dynamic_header(in, p.c, table, sizes);
        if (!complete && in.byte + 8 > avail) return kParseNeedMore; // (the header reader may have run off the head)
        if (!ok) return fpng::FPNG_DECODE_NOT_FPNG;
        // a table with codes for the reserved length symbols 286 / 287: the reference's 4-channel decoder gives them a meaning of its
        // own (fpng_decode.cpp: inflate_rows) -- no fpng encoder writes such a table; the CPU decoder's
        if (sizes[286] | sizes[287]) return FPNG_AMD_DECODE_UNDECIDED;
        p.first_bit = in.bitpos();
        if (memo && (p.first_bit >> 3) < sizeof memo->bytes && (p.first_bit >> 3) < avail) {
            memo->bits = (uint32_t)p.first_bit, memo->chans = p.c;
            std::memcpy(memo->bytes, z, (size_t)(p.first_bit >> 3) + 1);
            std::memcpy(memo->sizes, sizes, 288);
        }
    }
    if (p.first_bit >= (uint64_t)(p.idat_len - 4) * 8) return fpng::FPNG_DECODE_NOT_FPNG;
    // a token has at least 2 bits and stands for at most 258 bytes: an IDAT this short cannot hold the image (checked before any
    // device memory is sized by the header's dimensions)
    if (total > (uint64_t)p.idat_len * 1032 + 258) return fpng::FPNG_DECODE_NOT_FPNG;
    return 0;
}

// parse one file that is wholly in host memory
int parse_host(const uint8_t *png, uint32_t size, Parsed &p, uint32_t *table, uint8_t sizes[288], HeaderMemo *memo = nullptr)
{
    p.status = fpng::parse::parse_container(png, size, p.w, p.h, p.c, p.idat_ofs, p.idat_len);
    if (p.status) return p.status;
    return plan_stream(png + p.idat_ofs + 8, size - (p.idat_ofs + 8), true, p, table, sizes, memo);
}

// A new epoch for dec_unfilter_kernel's look-back granules (never cleared: the epoch tells launches apart).  0 is skipped: it is the
// tag of granules that were zeroed and never written.
uint32_t next_epoch(fpng_amd_encoder *e)
{
    if (!(++e->dec_epoch & 0x3FFFFFFFu)) ++e->dec_epoch;
    return e->dec_epoch & 0x3FFFFFFFu;
}

int decode_files(fpng_amd_encoder *e, const fpng_amd_png *files, uint32_t n, uint32_t desired, fpng_amd_decode_result *results, bool device_data)
{
    if (!e || !files || !n || !results) return fail(FPNG_AMD_ERR_INVALID_ARG, "null/empty batch");
    if (desired != 3 && desired != 4) return fail(FPNG_AMD_ERR_INVALID_ARG, "desired_chans must be 3 or 4");
    HIP_TRY(hipSetDevice(e->device));
    int rc = drain(e);
    if (rc) return rc;
    using namespace fpng::parse;
    hipStream_t s = e->stream;
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
    const uint32_t resident = (uint32_t)std::max(cus, 1) * 3; // persistent workgroups of dec_sync_kernel's border rounds: their LDS lets three share a compute unit
    uint32_t max_rounds = kMaxRounds;
    if (const char *mr = getenv("FPNG_AMD_DECODE_MAX_ROUNDS")) max_rounds = (uint32_t)std::max(0, atoi(mr)); // (0: every dynamic file is left to the CPU decoder -- tests)
    static const bool trace_t = getenv("FPNG_AMD_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::memset(results, 0, (size_t)n * sizeof *results); // (every entry is defined on every way out)
    std::vector<Parsed> ps(n);
    std::vector<uint8_t> lut_keys;                // the code lengths (288 each) of the unique lookup tables (1-pass files share two): ONE upload,
                                                  // the tables themselves are built on the GPU (dec_build_lut_kernel)
    std::unordered_multimap<uint64_t, uint32_t> lut_index; // ... found by their hash
    uint32_t *const table = nullptr; // (no host-side lookup table: the header reader only checks the code)
    HeaderMemo memo;
    int prev_lut = -1; // the table of the last dynamic file (index into lut_keys)
    std::vector<DecJob> jobs;
    std::vector<uint32_t> job_file;
    size_t z_total = 0, win_total = 0, seg_total = 0;
    uint32_t sub_total = 0;

    // ---- device-resident files: their first and last bytes come back first (one round trip for the batch) ----
    if (device_data) {
        const size_t per = kHeadBytes + kTailBytes, refs = ((size_t)n * sizeof(DecFileRef) + 255) & ~(size_t)255;
        if ((rc = e->h_dec_fetch.ensure((size_t)n * per + refs)) || (rc = e->d_decode.ensure((size_t)n * per + refs))) return rc;
        DecFileRef *h_refs = (DecFileRef *)(e->h_dec_fetch.p + (size_t)n * per);
        for (uint32_t i = 0; i < n; i++) h_refs[i] = {(const uint8_t *)files[i].data, files[i].size, 0};
        // (one small upload, one gather kernel, one download: two copies per file cost ~10 us each)
        HIP_TRY(hipMemcpyAsync(e->d_decode.p + (size_t)n * per, h_refs, (size_t)n * sizeof(DecFileRef), hipMemcpyHostToDevice, s));
        launch_dec_fetch(s, (const DecFileRef *)(e->d_decode.p + (size_t)n * per), n, kHeadBytes, kTailBytes, e->d_decode.p);
        HIP_TRY(hipMemcpyAsync(e->h_dec_fetch.p, e->d_decode.p, (size_t)n * per, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (trace_t) fprintf(stderr, "[decode] +%.0f us: heads and tails of %u files are here\n", since(), n);
    }
    std::vector<uint8_t> whole; // a device-resident file the head and tail were not enough for
    for (uint32_t i = 0; i < n; i++) {
        Parsed &p = ps[i];
        fpng_amd_decode_result &r = results[i];
        std::memset(&r, 0, sizeof r);
        if (!files[i].data || !files[i].size) {
            r.status = fpng::FPNG_DECODE_INVALID_ARG;
            continue;
        }
        uint8_t sizes[288];
        int st;
        if (!device_data)
            st = parse_host((const uint8_t *)files[i].data, files[i].size, p, table, sizes, &memo);
        else {
            const uint8_t *buf = e->h_dec_fetch.p + (size_t)i * (kHeadBytes + kTailBytes);
            const uint32_t size = files[i].size, hl = std::min(size, kHeadBytes), tl = size > hl ? std::min(size - hl, kTailBytes) : 0u;
            const View v = {buf, hl, tl ? buf + kHeadBytes : nullptr, size - tl, size};
            st = p.status = parse_container_view(v, p.w, p.h, p.c, p.idat_ofs, p.idat_len);
            if (!st) st = (p.idat_ofs + 8 < hl) ? plan_stream(buf + p.idat_ofs + 8, hl - (p.idat_ofs + 8), hl == size, p, table, sizes, &memo) : kParseNeedMore;
            if (st == kParseNeedMore) { // unusual chunks, a stored file, ...: the whole file comes back
                whole.resize(size);
                HIP_TRY(hipMemcpy(whole.data(), files[i].data, size, hipMemcpyDeviceToHost));
                p = Parsed();
                st = parse_host(whole.data(), size, p, table, sizes, &memo);
            }
        }
        r.w = p.w, r.h = p.h, r.channels_in_file = p.c;
        r.status = st;
        if (p.status) continue; // the container's own status (geometry may be half known)
        const uint64_t need = (uint64_t)p.w * p.h * desired;
        if (need > UINT32_MAX) {
            r.status = fpng::FPNG_DECODE_FAILED_DIMENSIONS_TOO_LARGE;
            continue;
        }
        if (st) continue; // (reference :3131-3136: any stream problem is NOT_FPNG; or left to the CPU decoder)
        if (!files[i].d_pixels || files[i].pixels_cap < need) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "d_pixels / pixels_cap < w * h * desired_chans"); // (only files that will be written need room)
        if (!p.mode) {
            if (!max_rounds) {
                r.status = FPNG_AMD_DECODE_UNDECIDED;
                continue;
            }
            // (the table of the file in front -- every file of a 1-pass batch -- is found by one comparison; else by a hash over the code
            //  lengths, eight bytes a step: a byte a step was 0.25 us a file, most of what the host spent on a 1-pass file)
            if (prev_lut >= 0 && !std::memcmp(lut_keys.data() + (size_t)prev_lut * 288, sizes, 288)) p.lut = prev_lut;
            if (p.lut < 0) {
                uint64_t hsh = 1469598103934665603ull; // (FNV-1a style: a batch of 2-pass files has a table per file)
                for (int q = 0; q < 288; q += 8) {
                    uint64_t v;
                    std::memcpy(&v, sizes + q, 8);
                    hsh = (hsh ^ v) * 1099511628211ull;
                    hsh ^= hsh >> 29;
                }
                auto range = lut_index.equal_range(hsh);
                for (auto it = range.first; it != range.second && p.lut < 0; ++it)
                    if (!std::memcmp(lut_keys.data() + (size_t)it->second * 288, sizes, 288)) p.lut = (int)it->second;
                if (p.lut < 0) {
                    p.lut = (int)(lut_keys.size() / 288);
                    lut_index.emplace(hsh, (uint32_t)p.lut);
                    lut_keys.insert(lut_keys.end(), sizes, sizes + 288);
                }
            }
            prev_lut = p.lut;
        }
        DecJob j;
        std::memset(&j, 0, sizeof j);
        j.w = p.w, j.h = p.h, j.src_c = p.c, j.dst_c = desired, j.bpl = p.w * p.c;
        j.z_bytes = p.idat_len, j.first_bit = p.first_bit, j.end_limit_bit = (uint64_t)(p.idat_len - 4) * 8;
        j.mode = p.mode;
        j.out = files[i].d_pixels;
        j.sub_base = sub_total;
        if (!p.mode) {
            j.n_sub = (uint32_t)((j.end_limit_bit - j.first_bit + kSubBits - 1) / kSubBits);
            sub_total += (j.n_sub + kDecSubBlock - 1) / kDecSubBlock * kDecSubBlock; // whole workgroups per file
            // offsets into the shared scratch (pointers are patched once the buffers exist)
            j.win = (uint32_t *)(uintptr_t)win_total; // (words, patched below)
            win_total += (size_t)j.h * dec_col_blocks(j.w, j.src_c, j.dst_c) * dec::kWinWords;
            j.nseg = (p.h + kDecUnfRows - 1) / kDecUnfRows;
            j.segsum = (uint32_t *)(uintptr_t)seg_total;
            seg_total += (size_t)j.nseg * ((j.bpl + 3) / 4);
        }
        if (device_data) {
            const uintptr_t zr = (uintptr_t)files[i].data + p.idat_ofs + 8;
            j.z = (const uint8_t *)(zr & ~(uintptr_t)3);
            j.z_shift = (uint32_t)(zr & 3);
            j.z_bytes += j.z_shift, j.first_bit += 8 * j.z_shift, j.end_limit_bit += 8 * j.z_shift;
        } else {
            j.z = (const uint8_t *)(uintptr_t)z_total;
            z_total += ((size_t)p.idat_len + 16 + 15) & ~(size_t)15; // (a token's window reaches up to 8 bytes ahead)
        }
        jobs.push_back(j);
        job_file.push_back(i);
    }
    const uint32_t nj = (uint32_t)jobs.size();
    if (trace_t) fprintf(stderr, "[decode] +%.0f us: %u files parsed, %zu lookup tables wanted\n", since(), n, lut_keys.size() / 288);
    if (!nj) return FPNG_AMD_OK;

    // ---- device scratch: one encoder-owned buffer, carved up (kept between calls) ----
    uint8_t *d_z;
    uint32_t *d_win;
    uint32_t *d_status, *d_changed;
    unsigned long long *d_seg;
    uint64_t *d_block_off;
    DecBlockRec *d_recs;
    DecSubArrays d_sub;
    uint32_t *d_luts;
    uint8_t *d_keys;
    DecJob *d_jobs;
    uint8_t *d_plan;
    const size_t subs = std::max<size_t>(sub_total, 1), blocks = (subs + kDecSubBlock - 1) / kDecSubBlock;
    size_t setup_ofs = 0, setup_len = 0, setup_plan = 0; // job records, un-filter plans and the (zero) status words: adjacent in the scratch, ONE upload
    {
        size_t need = 0;
        auto carve = [&](size_t bytes) {
            const size_t o = need;
            need += (bytes + 255) & ~(size_t)255;
            return o;
        };
        // (the token records: dec::kRecRows rows of 64 8-byte entries per 64 subsequences -- 16 x the files' bytes, of which a gradient touches a fifth)
        const size_t o_z = carve(z_total + 64), o_win = carve(std::max<size_t>(win_total, 1) * 4), o_info = carve(subs * 4), o_bytes = carve(subs * 4),
                     o_rel = carve(subs * 4), o_last = carve(subs * 4), o_eob = carve(subs * 4), o_tok = carve((subs * (size_t)dec::kRecRows + 32 * dec::kRecLane) * 8),
                     o_recs = carve(blocks * sizeof(DecBlockRec)), o_boff = carve(blocks * 8),
                     o_luts = carve(std::max<size_t>(lut_keys.size() / 288, 1) * dec::kLutDwords * 4), o_keys = carve(std::max<size_t>(lut_keys.size(), 288)),
                     o_jobs = carve(nj * sizeof(DecJob)), o_plan = carve(((size_t)nj + kMaxGroups) * (sizeof(DecUnfPiece) + 8)), o_status = carve((2 * (size_t)nj + 1 + 2 * kMaxGroups) * 4);
        if ((rc = e->d_decode.ensure(need))) return rc;
        // the look-back granules of dec_unfilter_kernel: never cleared between calls -- every launch has its own epoch, and memory
        // that was just allocated (any bit pattern) is zeroed once
        if ((rc = e->d_dec_gran.ensure(std::max<size_t>(seg_total, 1)))) return rc;
        if (e->d_dec_gran.fresh) {
            HIP_TRY(hipMemsetAsync(e->d_dec_gran.p, 0, e->d_dec_gran.cap * 8, e->stream));
            e->d_dec_gran.fresh = false;
        }
        d_seg = e->d_dec_gran.p;
        uint8_t *base = e->d_decode.p;
        d_z = base + o_z, d_win = (uint32_t *)(base + o_win);
        d_sub.info = (uint32_t *)(base + o_info), d_sub.bytes = (uint32_t *)(base + o_bytes);
        d_sub.rel = (uint32_t *)(base + o_rel), d_sub.lastpx = (uint32_t *)(base + o_last), d_sub.eob = (uint32_t *)(base + o_eob), d_sub.tok = (uint64_t *)(base + o_tok);
        d_recs = (DecBlockRec *)(base + o_recs), d_block_off = (uint64_t *)(base + o_boff);
        // the windows' index: "no subsequence" until dec_subscan_kernel says otherwise (a stream that covers less than the image leaves holes)
        HIP_TRY(hipMemsetAsync(d_win, 0xFF, std::max<size_t>(win_total, 1) * 4, e->stream));
        d_luts = (uint32_t *)(base + o_luts), d_keys = base + o_keys, d_jobs = (DecJob *)(base + o_jobs), d_status = (uint32_t *)(base + o_status);
        d_plan = base + o_plan;
        setup_ofs = o_jobs, setup_plan = o_plan - o_jobs, setup_len = o_status + (2 * (size_t)nj + 1 + 2 * kMaxGroups) * 4 - o_jobs;
    }
    d_changed = d_status + nj; // (one word per group of files; 2 * nj + 1 + 2 * kMaxGroups words were carved out)
    uint32_t *d_eob = d_status + nj + kMaxGroups;
    uint32_t *d_multi = d_status + 2 * nj + kMaxGroups + 1; // (one word per group: launch_dec_sync)
    // the tables: from the encoder's cache when every one of this batch's is there; a batch of few distinct tables that are not
    // refills the cache (its tables are built in place); a batch of many (2-pass files: one each) builds them in the call's scratch
    const uint32_t n_luts = (uint32_t)(lut_keys.size() / 288);
    bool luts_cached = false;
    std::vector<uint32_t> lut_slot(n_luts, 0);
    const bool few_luts = n_luts && n_luts <= fpng_amd_encoder::kDecLutCache;
    if (few_luts) {
        if ((rc = e->d_lut_cache.ensure((size_t)fpng_amd_encoder::kDecLutCache * dec::kLutDwords))) return rc;
        if (e->d_lut_cache.fresh) e->lut_cache_n = 0, e->d_lut_cache.fresh = false;
        luts_cached = true;
        for (uint32_t q = 0; q < n_luts && luts_cached; q++) {
            bool hit = false;
            for (uint32_t c = 0; c < e->lut_cache_n && !hit; c++)
                if (!std::memcmp(e->lut_cache_keys[c], lut_keys.data() + (size_t)q * 288, 288)) lut_slot[q] = c, hit = true;
            luts_cached = hit;
        }
        if (!luts_cached) { // (re)fill: this batch's tables become the cache
            // The cache holds NOTHING until launch_dec_build_luts has been enqueued (further down): every error return between here
            // and there leaves it empty instead of naming tables that were never built.
            e->lut_cache_n = 0;
            for (uint32_t q = 0; q < n_luts; q++) std::memcpy(e->lut_cache_keys[q], lut_keys.data() + (size_t)q * 288, 288), lut_slot[q] = q;
            d_luts = e->d_lut_cache.p; // (built below, in place)
        }
    }
    for (uint32_t k = 0; k < nj; k++) {
        DecJob &j = jobs[k];
        const Parsed &p = ps[job_file[k]];
        if (!device_data) j.z = d_z + (size_t)(uintptr_t)j.z;
        if (!j.mode) {
            j.win = d_win + (size_t)(uintptr_t)j.win;
            j.segsum = (uint32_t *)(d_seg + (size_t)(uintptr_t)j.segsum);
            j.lut = (few_luts ? e->d_lut_cache.p + (size_t)lut_slot[p.lut] * dec::kLutDwords : d_luts + (size_t)p.lut * dec::kLutDwords);
        }
    }
    // ---- groups of files: while one group is decoded the next one's bytes are on their way (its own stream; from pageable
    //      memory an "asynchronous" copy keeps its caller busy for most of its duration, so a thread of its own issues them).
    //      Files that are in device memory already form one group. ----
    struct Group {
        uint32_t j0, j1, blk0, blk1;
        DecUnfPlan plan; // (device pointers)
    };
    std::vector<Group> groups;
    {
        // (file k goes to the group its middle byte falls into when the batch's bytes are cut into `want` equal parts)
        // Files in device memory are ONE group.  (Cutting them into 2..4 groups whose kernels alternate between two streams -- one
        // group's un-filter pass, memory-bound, under the next group's synchronisation and emit passes -- was measured in round 5: no
        // gain, 8 x 8K 2.33 -> 2.39 / 2.45 / 2.46 ms, profiles/r05_decode_groups.txt.)
        uint64_t total = 0, run = 0;
        for (uint32_t k = 0; k < nj; k++) total += jobs[k].z_bytes;
        const uint32_t want = (!device_data && z_total >= (8u << 20) && nj > 1) ? std::min<uint32_t>(kMaxGroups, nj) : 1u;
        auto close = [&](uint32_t j0, uint32_t j1) {
            Group g = {j0, j1, jobs[j0].sub_base / kDecSubBlock, (j1 < nj ? jobs[j1].sub_base : sub_total) / kDecSubBlock, {}};
            groups.push_back(g);
        };
        uint32_t j0 = 0, cur = 0;
        for (uint32_t k = 0; k < nj; k++) {
            const uint32_t gi = (uint32_t)std::min<uint64_t>(want - 1, (run + jobs[k].z_bytes / 2) * want / std::max<uint64_t>(total, 1));
            if (k > j0 && gi != cur) close(j0, k), j0 = k;
            cur = gi;
            run += jobs[k].z_bytes;
        }
        close(j0, nj);
    }
    const uint32_t ng = (uint32_t)groups.size();
    auto upload_group = [&](const Group &g, hipStream_t st) -> hipError_t {
        for (uint32_t k = g.j0; k < g.j1; k++) {
            const Parsed &p = ps[job_file[k]];
            const hipError_t err = hipMemcpyAsync((void *)jobs[k].z, (const uint8_t *)files[job_file[k]].data + p.idat_ofs + 8, p.idat_len, hipMemcpyHostToDevice, st);
            if (err != hipSuccess) return err;
        }
        return hipSuccess;
    };
    std::mutex mu;
    std::condition_variable cv;
    uint32_t issued = 0; // groups whose copies are enqueued (and whose event is recorded)
    hipError_t up_err = hipSuccess;
    struct Joiner { // (every way out of this function waits for the uploader first: it works on this frame's variables)
        Worker *w = nullptr;
        ~Joiner()
        {
            if (w) w->wait();
        }
    } joiner;
    const bool uploads = !device_data && ng > 1; // groups of host-resident files: their bytes go up on a stream of their own
    if (uploads) {
        if (!e->dec_up) HIP_TRY(create_copy_stream(&e->dec_up));
        for (uint32_t g = 0; g < ng; g++)
            if (!e->dec_ev[g]) HIP_TRY(hipEventCreateWithFlags(&e->dec_ev[g], hipEventDisableTiming));
        if (!e->workers) e->workers = new HostWorkers(); // (the encoder's copy threads, made once: host_workers.h)
        joiner.w = &e->workers->up;
        e->workers->up.start([&] {
            hipError_t err = hipSetDevice(e->device);
            for (uint32_t g = 0; g < ng; g++) {
                if (err == hipSuccess) err = upload_group(groups[g], e->dec_up);
                if (err == hipSuccess) err = hipEventRecord(e->dec_ev[g], e->dec_up);
                if (trace_t) fprintf(stderr, "[decode] +%.0f us: uploads of group %u issued\n", since(), g);
                std::lock_guard<std::mutex> lk(mu);
                if (err != hipSuccess) up_err = err;
                issued = g + 1;
                cv.notify_all();
            }
        });
    }
    if (!lut_keys.empty() && !luts_cached) {
        HIP_TRY(hipMemcpyAsync(d_keys, lut_keys.data(), lut_keys.size(), hipMemcpyHostToDevice, s));
        launch_dec_build_luts(s, d_keys, (uint32_t)(lut_keys.size() / 288), d_luts);
        if (few_luts) e->lut_cache_n = n_luts; // (the build is in the stream, in front of everything that will read the tables: now the cache names them)
    }
    // (four small uploads -- job records, plan pieces, plan words, cleared status words -- were four blit kernels with their dispatch
    //  gaps in front of the first decode kernel, ~7 us each: they go up as one block from pinned memory)
    if ((rc = e->h_dec_fetch.ensure(setup_len))) return rc;
    uint8_t *const h_setup = e->h_dec_fetch.p; // (the fetched heads and tails that lived here have been parsed)
    std::memset(h_setup, 0, setup_len);
    {   // dec_unfilter_kernel's work items per group of files, numbered segment by segment (decode.h: DecUnfPlan)
        DecUnfPiece *d_pieces = (DecUnfPiece *)d_plan;
        uint32_t *d_words = (uint32_t *)(d_pieces + nj + kMaxGroups);
        std::vector<DecUnfPiece> pieces;
        std::vector<uint32_t> words; // per group: cbpre (files + 1), then order (files)
        for (Group &g : groups) {
            std::vector<uint32_t> order;
            for (uint32_t q = g.j0; q < g.j1; q++)
                if (!jobs[q].mode) order.push_back(q - g.j0);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return jobs[g.j0 + a].nseg > jobs[g.j0 + b].nseg; });
            const uint32_t m = (uint32_t)order.size();
            const size_t w0 = words.size(), p0 = pieces.size();
            words.push_back(0);
            for (uint32_t k = 0; k < m; k++) words.push_back(words.back() + dec_col_blocks(jobs[g.j0 + order[k]].w, jobs[g.j0 + order[k]].src_c, jobs[g.j0 + order[k]].dst_c));
            words.insert(words.end(), order.begin(), order.end());
            uint32_t seg = 0, item = 0;
            for (uint32_t alive = m; alive >= 1; alive--) { // the alive-th file of the order is the next one to run out of rows
                const uint32_t end = jobs[g.j0 + order[alive - 1]].nseg;
                if (end > seg) {
                    pieces.push_back({item, seg, alive, 0});
                    item += (end - seg) * words[w0 + alive];
                    seg = end;
                }
            }
            g.plan.pieces = d_pieces + p0, g.plan.n_pieces = (uint32_t)(pieces.size() - p0), g.plan.total_items = item;
            g.plan.cbpre = d_words + w0, g.plan.order = d_words + w0 + m + 1, g.plan.n_files = m, g.plan.pad_ = 0;
        }
        if (!pieces.empty()) std::memcpy(h_setup + setup_plan, pieces.data(), pieces.size() * sizeof(DecUnfPiece));
        if (!words.empty()) std::memcpy(h_setup + setup_plan + ((size_t)nj + kMaxGroups) * sizeof(DecUnfPiece), words.data(), words.size() * 4);
    }
    std::memcpy(h_setup, jobs.data(), nj * sizeof(DecJob));
    HIP_TRY(hipMemcpyAsync(e->d_decode.p + setup_ofs, h_setup, setup_len, hipMemcpyHostToDevice, s));
    // profiling (fpng_amd_encoder_set_profiling): events around the kernels of the first group of files
    const bool prof = e->profiling;
    e->dec_prof_recorded = false;
    if (prof)
        for (hipEvent_t &ev : e->dec_prof_ev)
            if (!ev) HIP_TRY(hipEventCreate(&ev));
    auto stamp = [&](const Group &g, int k) -> hipError_t { return (prof && &g == groups.data()) ? hipEventRecord(e->dec_prof_ev[k], s) : hipSuccess; };
    auto finish_group = [&](const Group &g, hipStream_t s) -> hipError_t { // everything behind the synchronisation (every step of it is idempotent)
        const uint32_t nblk = g.blk1 - g.blk0;
        hipError_t pe = stamp(g, 1);
        if (pe != hipSuccess) return pe;
        if (nblk) {
            launch_dec_offsets(s, d_jobs, nj, g.blk0, nblk, sub_total, d_jobs + g.j0, g.j1 - g.j0, d_sub, d_recs, d_block_off, d_status, d_eob);
            if ((pe = stamp(g, 2)) != hipSuccess) return pe;
        } else if ((pe = stamp(g, 2)) != hipSuccess)
            return pe;
        if ((pe = stamp(g, 3)) != hipSuccess) return pe;
        bool any_stored = false;
        for (uint32_t k = g.j0; k < g.j1; k++) any_stored |= jobs[k].mode != 0;
        // (all groups run on one stream: two un-filter kernels never run at once -- each one's workgroups wait for lower-numbered ones
        //  of their own launch, and two sets of waiting workgroups could keep each other's predecessors off the compute units)
        const DecPlaced placed = {d_sub, d_block_off, d_eob + g.j0, 0xFFFFFFFFu};
        launch_dec_finish(s, d_jobs + g.j0, g.j1 - g.j0, g.plan, placed, d_status + g.j0, next_epoch(e), any_stored);
        if ((pe = stamp(g, 4)) != hipSuccess) return pe;
        if (prof && &g == groups.data()) e->dec_prof_recorded = true;
        return hipSuccess;
    };
    for (uint32_t gi = 0; gi < ng; gi++) {
        const Group &g = groups[gi];
        if (ng == 1) {
            if (!device_data) HIP_TRY(upload_group(g, s));
        } else if (uploads) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return issued > gi; });
            if (up_err != hipSuccess) return fail(FPNG_AMD_ERR_HIP, "upload of the files", up_err);
            lk.unlock();
            HIP_TRY(hipStreamWaitEvent(s, e->dec_ev[gi], 0));
        }
        if (trace_t) fprintf(stderr, "[decode] +%.0f us: group %u (files %u..%u, %u workgroups) starts\n", since(), gi, g.j0, g.j1, g.blk1 - g.blk0);
        const uint32_t nblk = g.blk1 - g.blk0;
        HIP_TRY(stamp(g, 0));
        // round 0 settles every workgroup in itself; the borders between workgroups get kBorderRounds rounds launched blind (a
        // workgroup whose border holds leaves at once), and the chain check of dec_offsets_kernel says whether that was enough
        for (uint32_t r = 0; nblk && r <= std::min(kBorderRounds, max_rounds - 1); r++) launch_dec_sync(s, resident, d_jobs, nj, g.blk0, nblk, sub_total, r, d_sub, d_recs, d_changed + gi, d_multi + gi);
        HIP_TRY(finish_group(g, s));
        if (trace_t) fprintf(stderr, "[decode] +%.0f us: group %u enqueued\n", since(), gi);
    }
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> status(nj);
    HIP_TRY(hipMemcpyAsync(status.data(), d_status, nj * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // groups with a file whose chain does not hold across some border yet (nothing of such a file was written): more rounds, in
    // fours, until one changes nothing -- a stream whose decoders stay out of step over whole workgroups (periodic content) needs
    // a round per border and is left to the CPU decoder beyond max_rounds -- then the rest of the pipeline again
    bool again = false;
    for (uint32_t gi = 0; gi < ng; gi++) {
        const Group &g = groups[gi];
        bool open = false;
        for (uint32_t k = g.j0; k < g.j1; k++) open |= (status[k] & kDecNotConverged) != 0;
        const uint32_t nblk = g.blk1 - g.blk0;
        if (!open || !nblk) continue;
        again = true;
        uint32_t r = std::min(kBorderRounds, max_rounds - 1);
        for (bool settled = false; !settled && r + 1 < max_rounds;) {
            uint32_t changed = 0;
            for (int k = 0; k < 4; k++) {
                r++;
                if (k == 3) HIP_TRY(hipMemsetAsync(d_changed + gi, 0, 4, s)); // (only the last of the four is asked)
                launch_dec_sync(s, resident, d_jobs, nj, g.blk0, nblk, sub_total, r, d_sub, d_recs, d_changed + gi, d_multi + gi);
            }
            HIP_TRY(hipMemcpyAsync(&changed, d_changed + gi, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            settled = !changed;
        }
        if (trace_t) fprintf(stderr, "[decode] +%.0f us: group %u needed %u rounds\n", since(), gi, r + 1);
        HIP_TRY(hipMemsetAsync(d_status + g.j0, 0, (g.j1 - g.j0) * 4, s));
        HIP_TRY(finish_group(g, s));
    }
    if (again) {
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(status.data(), d_status, nj * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (trace_t) fprintf(stderr, "[decode] +%.0f us: done\n", since());
#ifdef FPNG_DEC_TILE_TIMING
    if (const char *tp = getenv("FPNG_AMD_TILE_TIMES")) dec_dump_tile_times(tp, groups[0].plan.total_items);
#endif
#ifdef FPNG_DEC_SYNC_TIMING
    if (const char *tp = getenv("FPNG_AMD_SYNC_TIMES")) dec_dump_sync_times(tp, groups[0].blk1 - groups[0].blk0);
#endif
    static const bool trace = getenv("FPNG_AMD_TRACE") != nullptr;
    for (uint32_t k = 0; k < nj; k++) {
        if (trace)
            fprintf(stderr, "[decode] file %u: %ux%ux%u mode %u, %u subsequences, first bit %llu, device status 0x%x\n", job_file[k], jobs[k].w, jobs[k].h,
                    jobs[k].src_c, jobs[k].mode, jobs[k].n_sub, (unsigned long long)jobs[k].first_bit, status[k]);
        int32_t &st = results[job_file[k]].status;
        if (jobs[k].mode) {
            if (status[k] & kDecStoredOdd) { // not the usual stored layout after all: the whole file, on the host (check_stored() is the rule)
                st = FPNG_AMD_DECODE_UNDECIDED;
                if (device_data) {
                    const Parsed &p = ps[job_file[k]];
                    whole.resize(p.idat_len);
                    HIP_TRY(hipMemcpy(whole.data(), (const uint8_t *)files[job_file[k]].data + p.idat_ofs + 8, p.idat_len, hipMemcpyDeviceToHost));
                    if (check_stored(whole.data(), p.idat_len, p.idat_len, p.w, p.h, p.c) == 1) st = fpng::FPNG_DECODE_NOT_FPNG;
                }
            }
            continue;
        }
        if (status[k] & (kDecNotConverged | kDecStalled)) // (nothing else is known then: "invalid" may be a speculative decode's)
            st = FPNG_AMD_DECODE_UNDECIDED;
        else if (status[k] & (kDecBadStream | kDecBadFilter))
            st = fpng::FPNG_DECODE_NOT_FPNG;
        else if (!(status[k] & kDecSawEob))
            st = fpng::FPNG_DECODE_NOT_FPNG; // the stream never ended
    }
    return FPNG_AMD_OK;
}

} // namespace

extern "C" int fpng_amd_decode_last_phase_ms(fpng_amd_encoder *e, float ms[FPNG_AMD_NUM_DECODE_PHASES])
{
    if (!e || !ms) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < FPNG_AMD_NUM_DECODE_PHASES; i++) ms[i] = 0.f;
    if (!e->dec_prof_recorded) return FPNG_AMD_OK;
    HIP_TRY(hipEventSynchronize(e->dec_prof_ev[4]));
    for (int i = 0; i < FPNG_AMD_NUM_DECODE_PHASES; i++) HIP_TRY(hipEventElapsedTime(&ms[i], e->dec_prof_ev[i], e->dec_prof_ev[i + 1]));
    return FPNG_AMD_OK;
}

extern "C" int fpng_amd_decode_batch(fpng_amd_encoder *e, const fpng_amd_png *files, uint32_t n, uint32_t desired, fpng_amd_decode_result *results)
{
    return decode_files(e, files, n, desired, results, false);
}

extern "C" int fpng_amd_decode_batch_device(fpng_amd_encoder *e, const fpng_amd_png *files, uint32_t n, uint32_t desired, fpng_amd_decode_result *results)
{
    return decode_files(e, files, n, desired, results, true);
}

namespace {

// fpng_amd_decode_host for a LARGE compressed file, streamed: the IDAT goes up in pieces (the encoder's uploader thread), every
// piece is synchronised, placed (dec_offsets_range_kernel carries the byte count from piece to piece) and decoded as soon as it has
// arrived, the rows that are complete get their Up filter undone and go down (the downloader thread) while later pieces are still
// on their way up: upload, decode and download overlap instead of following one another (8K RGBA: 58 MB up + 133 MB down).
// `out`: the caller's w * h * desired bytes.  *redo: the file's token boundaries did not settle in the rounds launched here --
// the caller goes through fpng_amd_decode_batch, which adds rounds.
int decode_host_streamed(fpng_amd_encoder *e, const uint8_t *png, const Parsed &p, const uint32_t *table, const uint8_t *sizes, uint32_t desired, uint8_t *out,
                         fpng_amd_decode_result *result, bool *redo)
{
    *redo = false;
    int rc;
    if ((rc = ensure_copy_streams(e))) return rc;
    hipStream_t s = e->stream, s_up = e->host.up, s_down = e->host.down;
    // The Up filter's undoing of a piece's rows runs on a stream of its own (a lane's: the lanes are drained), next to the following
    // piece's synchronisation and decode instead of behind them: the rows go down one piece earlier.
    hipStream_t s_unf = e->lane_stream[0] ? e->lane_stream[0] : s;
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
    const uint32_t resident = (uint32_t)std::max(cus, 1) * 3;
    std::vector<uint32_t> lut(dec::kLutDwords);
    build_multi_lut(table, sizes, lut.data());
    DecJob j;
    std::memset(&j, 0, sizeof j);
    j.w = p.w, j.h = p.h, j.src_c = p.c, j.dst_c = desired, j.bpl = p.w * p.c;
    j.z_bytes = p.idat_len, j.first_bit = p.first_bit, j.end_limit_bit = (uint64_t)(p.idat_len - 4) * 8;
    j.n_sub = (uint32_t)((j.end_limit_bit - j.first_bit + kSubBits - 1) / kSubBits);
    j.nseg = (p.h + kDecUnfRows - 1) / kDecUnfRows;
    const uint32_t n_blocks = (j.n_sub + kDecSubBlock - 1) / kDecSubBlock, sub_total = n_blocks * kDecSubBlock;
    const size_t col_blocks = dec_col_blocks(j.w, j.src_c, j.dst_c), os = (size_t)p.w * desired;
    // ---- scratch ----
    uint8_t *d_z;
    DecSubArrays d_sub;
    DecBlockRec *d_recs;
    uint64_t *d_block_off;
    uint32_t *d_lut, *d_words, *d_status, *d_eob, *d_win;
    DecJob *d_job;
    DecUnfPiece *d_piece;
    DecCarry *d_carry;
    {
        size_t need = 0;
        auto carve = [&](size_t bytes) {
            const size_t o = need;
            need += (bytes + 255) & ~(size_t)255;
            return o;
        };
        const size_t n_win = (size_t)j.h * col_blocks * dec::kWinWords;
        const size_t o_z = carve((size_t)p.idat_len + 80), o_win = carve(n_win * 4), o_info = carve((size_t)sub_total * 4), o_bytes = carve((size_t)sub_total * 4),
                     o_rel = carve((size_t)sub_total * 4), o_last = carve((size_t)sub_total * 4), o_eob = carve((size_t)sub_total * 4),
                     o_tok = carve(((size_t)sub_total * dec::kRecRows + 32 * dec::kRecLane) * 8), o_recs = carve(n_blocks * sizeof(DecBlockRec)),
                     o_boff = carve((size_t)n_blocks * 8), o_lut = carve(dec::kLutDwords * 4), o_job = carve(sizeof(DecJob)), o_small = carve(256);
        if ((rc = e->d_decode.ensure(need))) return rc;
        if ((rc = e->d_dec_gran.ensure(std::max<size_t>((size_t)j.nseg * ((j.bpl + 3) / 4), 1)))) return rc;
        if (e->d_dec_gran.fresh) {
            HIP_TRY(hipMemsetAsync(e->d_dec_gran.p, 0, e->d_dec_gran.cap * 8, s));
            e->d_dec_gran.fresh = false;
        }
        uint8_t *base = e->d_decode.p;
        d_z = base + o_z, d_win = (uint32_t *)(base + o_win);
        d_sub.info = (uint32_t *)(base + o_info), d_sub.bytes = (uint32_t *)(base + o_bytes);
        d_sub.rel = (uint32_t *)(base + o_rel), d_sub.lastpx = (uint32_t *)(base + o_last), d_sub.eob = (uint32_t *)(base + o_eob), d_sub.tok = (uint64_t *)(base + o_tok);
        d_recs = (DecBlockRec *)(base + o_recs), d_block_off = (uint64_t *)(base + o_boff), d_lut = (uint32_t *)(base + o_lut), d_job = (DecJob *)(base + o_job);
        HIP_TRY(hipMemsetAsync(d_win, 0xFF, n_win * 4, s)); // ("no subsequence": decode_files())
        uint8_t *sm = base + o_small; // status, eob index | carry | unfilter piece | cbpre[2], order[1]
        d_status = (uint32_t *)sm, d_eob = d_status + 1, d_carry = (DecCarry *)(sm + 16), d_piece = (DecUnfPiece *)(sm + 32), d_words = (uint32_t *)(sm + 48);
    }
    j.z = d_z, j.win = d_win, j.lut = d_lut, j.out = e->d_stage_in.p, j.segsum = (uint32_t *)e->d_dec_gran.p;
    struct Small { // (one upload for the few words the kernels start from)
        uint32_t status, eob;
        uint32_t pad0[2];
        DecCarry carry;
        DecUnfPiece piece;
        uint32_t cbpre[2], order[1];
    } small = {0, j.n_sub, {0, 0}, {0, 0, 0}, {0, 0, 1, 0}, {0, (uint32_t)col_blocks}, {0}};
    static_assert(sizeof(Small) <= 256 && offsetof(Small, carry) == 16 && offsetof(Small, piece) == 32 && offsetof(Small, cbpre) == 48, "layout of the small words");
    HIP_TRY(hipMemcpyAsync(d_status, &small, sizeof small, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_lut, lut.data(), dec::kLutDwords * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_job, &j, sizeof j, hipMemcpyHostToDevice, s));
    DecUnfPlan plan;
    plan.pieces = d_piece, plan.cbpre = d_words, plan.order = d_words + 2, plan.n_pieces = 1, plan.total_items = j.nseg * (uint32_t)col_blocks, plan.n_files = 1, plan.pad_ = 0;
    const uint32_t epoch = next_epoch(e); // (one epoch for all of this file's unfilter launches: later segments look back at earlier launches' sums)
    // ---- pieces: whole blocks of subsequences; a piece's kernels read up to 64 bytes behind its last block (a token's window, the pad) ----
    constexpr uint32_t kMaxPieces = 16;
    // (a piece costs ~100 us of launches and a host round trip: pieces of 6 MiB, but at least four of them -- measured on one box,
    //  2 / 3 / 4 / 6 MiB pieces: 8K RGBA 3.06 / 3.05 / 3.01 / 2.99 ms, an 11 MP photograph 2.46 / 1.90 / 1.53 / 1.38, 4K RGBA 1.03 / 0.95 / 1.02 / 1.06)
    const uint32_t np_want = std::max(4u, p.idat_len / (6u << 20));
    const uint32_t np = std::max(1u, std::min<uint32_t>({kMaxPieces, n_blocks, np_want}));
    // The first pieces are small -- a quarter, then half a share: the first rows are on their way down after 1 MiB instead of 4,
    // and the download, which takes longer than everything else together, starts that much earlier.
    constexpr bool ramp = true;
    uint32_t blk_end[kMaxPieces], byte_end[kMaxPieces];
    for (uint32_t k = 0; k < np; k++) {
        // shares: 1/4, 1/2, 1, 1, ... of (np - 1.25) equal ones
        const double done = (ramp && np >= 4) ? (k == 0 ? 0.25 : (k == 1 ? 0.75 : (double)k - 0.25)) / ((double)np - 1.25) : (double)(k + 1) / np;
        blk_end[k] = k + 1 == np ? n_blocks : std::min<uint32_t>(n_blocks, std::max<uint32_t>(k + 1, (uint32_t)(n_blocks * done)));
        const uint64_t end_bit = j.first_bit + (uint64_t)blk_end[k] * kDecSubBlock * kSubBits;
        byte_end[k] = k + 1 == np ? p.idat_len : (uint32_t)std::min<uint64_t>(p.idat_len, (end_bit >> 3) + 64);
    }
    hipEvent_t ev_up[kMaxPieces], ev_carry[kMaxPieces], ev_rows[kMaxPieces];
    for (uint32_t k = 0; k < np; k++) {
        if (!e->dec_ev[k]) HIP_TRY(hipEventCreateWithFlags(&e->dec_ev[k], hipEventDisableTiming));
        if (!e->dec_ev2[k]) HIP_TRY(hipEventCreateWithFlags(&e->dec_ev2[k], hipEventDisableTiming));
        if (!e->dec_ev3[k]) HIP_TRY(hipEventCreateWithFlags(&e->dec_ev3[k], hipEventDisableTiming));
        ev_up[k] = e->dec_ev[k], ev_carry[k] = e->dec_ev2[k], ev_rows[k] = e->dec_ev3[k];
    }
    if ((rc = e->h_dec_fetch.ensure(kMaxPieces * sizeof(DecCarry) + 64))) return rc;
    DecCarry *h_carry = (DecCarry *)e->h_dec_fetch.p;
    uint32_t *h_status = (uint32_t *)(h_carry + kMaxPieces);
    if (!e->workers) e->workers = new HostWorkers();
    std::mutex mu;
    std::condition_variable cv;
    uint32_t issued = 0, rows_ready = 0; // pieces whose upload is enqueued; pieces whose finished rows may go down
    uint32_t row_end[kMaxPieces] = {};   // rows [row_end[k - 1], row_end[k]) are complete behind piece k's unfilter launch
    hipError_t copy_err = hipSuccess;
    bool stop = false;
    const uint8_t *zsrc = png + p.idat_ofs + 8;
    struct Joiner { // (every way out waits for the copy threads: they work on this frame's variables)
        Worker &a, &b;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &stop;
        ~Joiner()
        {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            a.wait(), b.wait();
        }
    } joiner{e->workers->up, e->workers->down, mu, cv, stop};
    e->workers->up.start([&] {
        hipError_t err = hipSetDevice(e->device);
        for (uint32_t k = 0; k < np; k++) {
            const uint32_t from = k ? byte_end[k - 1] : 0;
            if (err == hipSuccess && byte_end[k] > from) err = hipMemcpyAsync(d_z + from, zsrc + from, byte_end[k] - from, hipMemcpyHostToDevice, s_up);
            if (err == hipSuccess) err = hipEventRecord(ev_up[k], s_up);
            std::lock_guard<std::mutex> lk(mu);
            if (err != hipSuccess) copy_err = err;
            issued = k + 1;
            cv.notify_all();
        }
    });
    e->workers->down.start([&] {
        hipError_t err = hipSetDevice(e->device);
        for (uint32_t k = 0; k < np; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return rows_ready > k || stop; });
                if (rows_ready <= k) return;
            }
            const uint32_t r0 = k ? row_end[k - 1] : 0, r1 = row_end[k];
            if (err == hipSuccess && r1 > r0) {
                err = hipStreamWaitEvent(s_down, ev_rows[k], 0);
                if (err == hipSuccess) err = hipMemcpyAsync(out + (size_t)r0 * os, e->d_stage_in.p + (size_t)r0 * os, (size_t)(r1 - r0) * os, hipMemcpyDeviceToHost, s_down);
            }
            if (err != hipSuccess) {
                std::lock_guard<std::mutex> lk(mu);
                copy_err = err;
            }
        }
        if (err == hipSuccess) err = hipStreamSynchronize(s_down);
        if (err != hipSuccess) {
            std::lock_guard<std::mutex> lk(mu);
            copy_err = err;
        }
    });
    // ---- the calling thread: piece k's kernels are enqueued, then piece k - 1 is closed (its byte count read, the rows it completed
    //      sent through the Up filter's undoing and handed to the downloader): the GPU always has the next piece's kernels queued ----
    uint32_t segs_done = 0;
    for (uint32_t k = 0; k <= np; k++) {
        if (k < np) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return issued > k; });
                if (copy_err != hipSuccess) return fail(FPNG_AMD_ERR_HIP, "upload of the file", copy_err);
            }
            HIP_TRY(hipStreamWaitEvent(s, ev_up[k], 0));
            const uint32_t a = k ? blk_end[k - 1] : 0, b = blk_end[k];
            if (b > a) {
                for (uint32_t r = 0; r <= kBorderRounds; r++) launch_dec_sync(s, resident, d_job, 1, a, b - a, sub_total, r, d_sub, d_recs, d_status + 2, d_status + 3);
                launch_dec_offsets_range(s, d_job, 0, a, b, k + 1 == np, sub_total, d_sub, d_recs, d_block_off, d_status, d_eob, d_carry);
            }
            HIP_TRY(hipMemcpyAsync(&h_carry[k], d_carry, sizeof(DecCarry), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(ev_carry[k], s));
        }
        if (k >= 1) {
            const uint32_t q = k - 1;
            HIP_TRY(hipEventSynchronize(ev_carry[q]));
            const uint64_t bytes = h_carry[q].bytes;
            uint32_t rows = (uint32_t)std::min<uint64_t>(bytes / ((uint64_t)j.bpl + 1), p.h);
            uint32_t segs = rows / kDecUnfRows; // whole segments only -- or, behind the last piece, everything
            if (q + 1 == np) segs = j.nseg, rows = p.h;
            else rows = segs * kDecUnfRows;
            if (s_unf != s) HIP_TRY(hipStreamWaitEvent(s_unf, ev_carry[q], 0)); // (piece q's rows are in the records of the subsequences placed so far; the small words went up in front of piece 0)
            // (the rows of these segments lie in the output of the subsequences placed so far: the tiles' walks stop there -- what the
            //  next piece's kernels are writing behind it meanwhile is not theirs to read)
            const DecPlaced placed = {d_sub, d_block_off, d_eob, blk_end[q] * kDecSubBlock};
            if (segs > segs_done) launch_dec_unfilter(s_unf, d_job, plan, placed, segs_done * (uint32_t)col_blocks, (segs - segs_done) * (uint32_t)col_blocks, d_status, epoch, s_unf != s);
            segs_done = std::max(segs_done, segs);
            HIP_TRY(hipEventRecord(ev_rows[q], s_unf));
            {
                std::lock_guard<std::mutex> lk(mu);
                row_end[q] = std::max(rows, q ? row_end[q - 1] : 0u);
                rows_ready = q + 1;
            }
            cv.notify_all();
        }
    }
    HIP_TRY(hipGetLastError());
    if (s_unf != s) HIP_TRY(hipStreamWaitEvent(s, ev_rows[np - 1], 0)); // (the last rows' filter literals are part of the status)
    HIP_TRY(hipMemcpyAsync(h_status, d_status, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    e->workers->down.wait();
    if (copy_err != hipSuccess) return fail(FPNG_AMD_ERR_HIP, "copies of the streamed decode", copy_err);
    const uint32_t st = *h_status;
    if (st & (kDecNotConverged | kDecStalled)) {
        *redo = true;
        return FPNG_AMD_OK;
    }
    if ((st & (kDecBadStream | kDecBadFilter)) || !(st & kDecSawEob)) result->status = fpng::FPNG_DECODE_NOT_FPNG;
    return FPNG_AMD_OK;
}

} // namespace

// One host-resident file to host pixels: what fpng::fpng_decode_memory() does for large images (fpng_decode.cpp).  The pixels land in
// the encoder's staging buffer and go down in one copy into memory obtained from `reserve` (asked once the file is known to decode).
extern "C" int fpng_amd_decode_host(fpng_amd_encoder *e, const void *png, uint32_t size, uint32_t desired, fpng_amd_reserve_fn reserve, void *user,
                                    fpng_amd_decode_result *result)
{
    if (!e || !result || !reserve) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    if (desired != 3 && desired != 4) return fail(FPNG_AMD_ERR_INVALID_ARG, "desired_chans must be 3 or 4");
    std::memset(result, 0, sizeof *result);
    if (!png || !size) {
        result->status = fpng::FPNG_DECODE_INVALID_ARG;
        return FPNG_AMD_OK;
    }
    uint32_t w = 0, h = 0, c = 0, idat_ofs = 0, idat_len = 0;
    const int st = fpng::parse::parse_container((const uint8_t *)png, size, w, h, c, idat_ofs, idat_len);
    result->w = w, result->h = h, result->channels_in_file = c;
    if (st) {
        result->status = st;
        return FPNG_AMD_OK;
    }
    const uint64_t need = (uint64_t)w * h * desired;
    if (need > UINT32_MAX) {
        result->status = fpng::FPNG_DECODE_FAILED_DIMENSIONS_TOO_LARGE;
        return FPNG_AMD_OK;
    }
    // the stream's shape first: a header that promises more pixels than the IDAT can hold must not size any device memory
    Parsed sp;
    sp.w = w, sp.h = h, sp.c = c, sp.idat_ofs = idat_ofs, sp.idat_len = idat_len;
    static thread_local uint32_t stable[1u << fpng::parse::kTableBits];
    uint8_t ssizes[288];
    {
        const int ss = plan_stream((const uint8_t *)png + idat_ofs + 8, size - (idat_ofs + 8), true, sp, stable, ssizes);
        if (ss) {
            result->status = ss;
            return FPNG_AMD_OK;
        }
    }
    HIP_TRY(hipSetDevice(e->device));
    int rc = drain(e);
    if (rc) return rc;
    if ((rc = e->d_stage_in.ensure((size_t)need + 16))) return rc;
    // large compressed files: upload, decode and download overlapped (FPNG_AMD_DECODE_STREAM=0: one after the other)
    static const bool stream_ok = [] {
        const char *v = getenv("FPNG_AMD_DECODE_STREAM");
        return !(v && v[0] == '0');
    }();
    if (stream_ok && sp.mode == 0 && idat_len >= (8u << 20) && getenv("FPNG_AMD_DECODE_MAX_ROUNDS") == nullptr) {
        uint8_t *out = reserve(user, (size_t)need);
        if (!out) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "no room for the pixels");
        bool redo = false;
        if ((rc = decode_host_streamed(e, (const uint8_t *)png, sp, stable, ssizes, desired, out, result, &redo))) return rc;
        if (!redo) return FPNG_AMD_OK;
    }
    fpng_amd_png f;
    std::memset(&f, 0, sizeof f);
    f.data = png, f.size = size, f.d_pixels = e->d_stage_in.p, f.pixels_cap = e->d_stage_in.cap;
    if ((rc = fpng_amd_decode_batch(e, &f, 1, desired, result))) return rc;
    if (result->status) return FPNG_AMD_OK;
    uint8_t *out = reserve(user, (size_t)need); // (the same memory again if the streamed attempt above asked already)
    if (!out) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "no room for the pixels");
    HIP_TRY(hipMemcpy(out, e->d_stage_in.p, (size_t)need, hipMemcpyDeviceToHost));
    return FPNG_AMD_OK;
}

// What fpng_amd_decode_batch() prepares on the host for one file, without a GPU (tests hold a model of the kernels against it):
// container status, geometry, stored or dynamic, the token stream's first bit and its limit, the kernels' lookup table.
extern "C" int fpng_amd_decode_plan(const void *png_, uint32_t size, fpng_amd_decode_result *result, uint32_t *mode, uint32_t *idat_ofs, uint32_t *idat_len,
                                    uint64_t *first_bit, uint64_t *end_limit_bit, uint32_t *lut)
{
    if (!png_ || !size || !result || !mode || !idat_ofs || !idat_len || !first_bit || !end_limit_bit || !lut) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    const uint8_t *png = (const uint8_t *)png_;
    std::memset(result, 0, sizeof *result);
    Parsed p;
    static thread_local uint32_t table[1u << fpng::parse::kTableBits];
    uint8_t sizes[288];
    result->status = parse_host(png, size, p, table, sizes);
    result->w = p.w, result->h = p.h, result->channels_in_file = p.c;
    *mode = 0, *idat_ofs = p.idat_ofs, *idat_len = p.idat_len, *first_bit = 0, *end_limit_bit = 0;
    if (result->status) return FPNG_AMD_OK;
    *mode = p.mode;
    if (!p.mode) {
        *first_bit = p.first_bit, *end_limit_bit = (uint64_t)(p.idat_len - 4) * 8;
        build_multi_lut(table, sizes, lut);
    }
    return FPNG_AMD_OK;
}
