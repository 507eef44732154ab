// decode_api.cpp -- host side of the GPU batch decoder (decode.hip): what the reference's fpng_decode_memory does in front of
// its pixel loops (reference src/fpng.cpp:2904-3222: container walk, IDAT checks, block type, dynamic header), then uploads and
// launches.  The parsing code is the CPU decoder's own (png_parse.h), so the status codes of damaged containers are the same.
#include "decode.h"
#include "encoder.h"
#include "png_parse.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace fpng_amd;

namespace {

struct Parsed {
    uint32_t w = 0, h = 0, c = 0, idat_ofs = 0, idat_len = 0;
    int status = 0;       // fpng::FPNG_DECODE_*
    uint32_t mode = 0;    // 0 dynamic, 1 stored
    uint64_t first_bit = 0;
    int lut = -1;         // index into the unique lookup tables
};

// the stored-block layout the reference accepts (src/fpng.cpp:2107-2207): block headers, sizes, filter bytes 0, exact end
bool check_stored(const uint8_t *z, uint32_t avail, uint32_t zlib_len, uint32_t w, uint32_t h, uint32_t c)
{
    const uint64_t stride = (uint64_t)w * c + 1, total = stride * h;
    uint64_t src = 2, got = 0;
    for (;;) {
        if (src + 5 > avail) return false;
        const bool final_block = z[src] & 1;
        if (((z[src] >> 1) & 3) != 0) return false;
        const uint32_t len = z[src + 1] | (z[src + 2] << 8), nlen = z[src + 3] | (z[src + 4] << 8);
        src += 5;
        if (len != (~nlen & 0xFFFF) || src + len > avail) return false;
        // the GPU copy assumes the encoder's layout: full 65535-byte blocks, then the rest
        if (!final_block && len != 65535) return false;
        for (uint64_t r = (got + stride - 1) / stride * stride; r < got + len; r += stride) // filter bytes inside this block
            if (z[src + (r - got)] != 0) return false;
        got += len;
        src += len;
        if (final_block) break;
    }
    return got == total && src + 4 == zlib_len;
}

} // namespace

extern "C" int fpng_amd_decode_batch(fpng_amd_encoder *e, const fpng_amd_png *files, uint32_t n, uint32_t desired, fpng_amd_decode_result *results)
{
    if (!e || !files || !n || !results) return fail(FPNG_AMD_ERR_INVALID_ARG, "null/empty batch");
    if (desired != 3 && desired != 4) return fail(FPNG_AMD_ERR_INVALID_ARG, "desired_chans must be 3 or 4");
    HIP_TRY(hipSetDevice(e->device));
    int rc = drain(e);
    if (rc) return rc;
    using namespace fpng::parse;
    std::vector<Parsed> ps(n);
    std::vector<std::vector<uint16_t>> luts;      // unique lookup tables (1-pass files share two)
    std::vector<std::vector<uint8_t>> lut_keys;   // the code lengths they were built from
    static thread_local uint32_t table[1u << kTableBits];
    std::vector<DecJob> jobs;
    std::vector<uint32_t> job_file;
    size_t z_total = 0, filt_total = 0, mask_total = 0;
    uint32_t sub_total = 0, max_rows = 0, max_bpl = 0;
    for (uint32_t i = 0; i < n; i++) {
        Parsed &p = ps[i];
        fpng_amd_decode_result &r = results[i];
        std::memset(&r, 0, sizeof r);
        const uint8_t *png = (const uint8_t *)files[i].data;
        if (!png || !files[i].size) {
            r.status = fpng::FPNG_DECODE_INVALID_ARG;
            continue;
        }
        p.status = parse_container(png, files[i].size, p.w, p.h, p.c, p.idat_ofs, p.idat_len);
        r.w = p.w, r.h = p.h, r.channels_in_file = p.c;
        if (p.status) {
            r.status = p.status;
            continue;
        }
        const uint64_t need = (uint64_t)p.w * p.h * desired;
        if (need > UINT32_MAX) {
            r.status = fpng::FPNG_DECODE_FAILED_DIMENSIONS_TOO_LARGE;
            continue;
        }
        if (!files[i].d_pixels || files[i].pixels_cap < need) return fail(FPNG_AMD_ERR_BUFFER_TOO_SMALL, "d_pixels / pixels_cap < w * h * desired_chans");
        const uint8_t *z = png + p.idat_ofs + 8;
        const uint32_t avail = files[i].size - (p.idat_ofs + 8);
        r.status = fpng::FPNG_DECODE_NOT_FPNG; // until proven otherwise (reference :3131-3136: any stream problem)
        if (p.idat_len < 7 || z[0] != 0x78 || z[1] != 0x01) continue;
        if ((z[2] & 6) == 0) {
            if (!check_stored(z, avail, p.idat_len, p.w, p.h, p.c)) continue;
            p.mode = 1;
        } else {
            Bits in = {z, avail, 2, 0, 0, false};
            if (in.get(1) != 1 || in.get(2) != 2) continue; // one final dynamic block
            uint8_t sizes[288];
            if (!read_dynamic_header(in, p.c, table, sizes)) continue;
            p.first_bit = in.bitpos();
            if (p.first_bit >= (uint64_t)(p.idat_len - 4) * 8) continue;
            for (size_t k = 0; k < lut_keys.size() && p.lut < 0; k++)
                if (!std::memcmp(lut_keys[k].data(), sizes, 288)) p.lut = (int)k;
            if (p.lut < 0) {
                p.lut = (int)luts.size();
                lut_keys.emplace_back(sizes, sizes + 288);
                luts.emplace_back(1u << kTableBits);
                for (uint32_t k = 0; k < (1u << kTableBits); k++) luts.back()[k] = (uint16_t)table[k]; // symbol | length << 9 fits 13 bits
            }
        }
        r.status = 0;
        DecJob j;
        std::memset(&j, 0, sizeof j);
        j.w = p.w, j.h = p.h, j.src_c = p.c, j.dst_c = desired, j.bpl = p.w * p.c;
        j.z_bytes = p.idat_len, j.first_bit = p.first_bit, j.end_limit_bit = (uint64_t)(p.idat_len - 4) * 8;
        j.mode = p.mode;
        j.out = files[i].d_pixels;
        j.sub_base = sub_total;
        if (!p.mode) {
            j.n_sub = (uint32_t)((j.end_limit_bit - j.first_bit + kSubBits - 1) / kSubBits);
            sub_total += (j.n_sub + 255u) & ~255u; // whole workgroups per file
            // offsets into the shared scratch (pointers are patched once the buffers exist)
            j.filt = (uint8_t *)(uintptr_t)filt_total;
            j.runmask = (uint32_t *)(uintptr_t)mask_total;
            j.fstride = ((j.bpl + 3u) & ~3u) + 4u;
            filt_total += ((size_t)j.fstride * j.h + 15) & ~(size_t)15;
            mask_total += (size_t)((p.w + 31) / 32) * p.h;
            max_rows = std::max(max_rows, p.h), max_bpl = std::max(max_bpl, j.bpl);
        }
        j.z = (const uint8_t *)(uintptr_t)z_total;
        z_total += ((size_t)p.idat_len + 16 + 15) & ~(size_t)15; // (the bit reader looks up to 8 bytes ahead)
        jobs.push_back(j);
        job_file.push_back(i);
    }
    const uint32_t nj = (uint32_t)jobs.size();
    if (!nj) return FPNG_AMD_OK;

    // ---- device scratch: one encoder-owned buffer, carved up (kept between calls) ----
    uint8_t *d_z, *d_filt;
    uint32_t *d_mask, *d_bytes, *d_flags, *d_status, *d_changed;
    uint64_t *d_start, *d_end, *d_block_off;
    DecBlockRec *d_recs;
    uint16_t *d_luts;
    DecJob *d_jobs;
    const size_t subs = std::max<size_t>(sub_total, 1), blocks = (subs + 255) / 256;
    {
        size_t need = 0;
        auto carve = [&](size_t bytes) {
            const size_t o = need;
            need += (bytes + 255) & ~(size_t)255;
            return o;
        };
        const size_t o_z = carve(z_total + 64), o_filt = carve(filt_total), o_mask = carve(mask_total * 4), o_bytes = carve(subs * 4), o_flags = carve(subs * 4),
                     o_start = carve(subs * 8), o_end = carve(subs * 8), o_recs = carve(blocks * sizeof(DecBlockRec)), o_boff = carve(blocks * 8),
                     o_luts = carve(std::max<size_t>(luts.size(), 1) * 8192), o_jobs = carve(nj * sizeof(DecJob)), o_status = carve((2 * (size_t)nj + 1) * 4);
        if ((rc = e->d_decode.ensure(need))) return rc;
        uint8_t *base = e->d_decode.p;
        d_z = base + o_z, d_filt = base + o_filt, d_mask = (uint32_t *)(base + o_mask), d_bytes = (uint32_t *)(base + o_bytes);
        d_flags = (uint32_t *)(base + o_flags), d_start = (uint64_t *)(base + o_start), d_end = (uint64_t *)(base + o_end);
        d_recs = (DecBlockRec *)(base + o_recs), d_block_off = (uint64_t *)(base + o_boff);
        d_luts = (uint16_t *)(base + o_luts), d_jobs = (DecJob *)(base + o_jobs), d_status = (uint32_t *)(base + o_status);
    }
    d_changed = d_status + nj;
    uint32_t *d_eob = d_status + nj + 1;
    hipStream_t s = e->stream;
    for (uint32_t k = 0; k < nj; k++) {
        DecJob &j = jobs[k];
        const fpng_amd_png &f = files[job_file[k]];
        const Parsed &p = ps[job_file[k]];
        const size_t zo = (size_t)(uintptr_t)j.z;
        HIP_TRY(hipMemcpyAsync(d_z + zo, (const uint8_t *)f.data + p.idat_ofs + 8, p.idat_len, hipMemcpyHostToDevice, s));
        j.z = j.z_aligned = d_z + zo;
        if (!j.mode) {
            j.filt = d_filt + (size_t)(uintptr_t)j.filt;
            j.runmask = d_mask + (size_t)(uintptr_t)j.runmask;
            j.lut = d_luts + (size_t)p.lut * 4096;
        }
    }
    for (size_t k = 0; k < luts.size(); k++) HIP_TRY(hipMemcpyAsync(d_luts + k * 4096, luts[k].data(), 8192, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), nj * sizeof(DecJob), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(d_mask, 0, mask_total * 4, s));
    HIP_TRY(hipMemsetAsync(d_status, 0, (2 * nj + 1) * 4, s));
    if (sub_total) {
        // the speculative round, then synchronisation rounds in groups of four until a round changes nothing.  Typical files
        // settle in one or two rounds; nearly incompressible ones (codes of almost equal length do not re-synchronise) need up
        // to one round per subsequence: those are left to the CPU decoder beyond kMaxRounds
        constexpr uint32_t kMaxRounds = 64;
        uint32_t r = 0;
        launch_dec_sync(s, d_jobs, nj, sub_total, 0, d_start, d_end, d_bytes, d_flags, d_changed);
        for (bool settled = false; !settled && r < kMaxRounds;) {
            uint32_t changed = 0;
            for (int k = 0; k < 4; k++) {
                r++;
                if (k == 3) HIP_TRY(hipMemsetAsync(d_changed, 0, 4, s)); // (only the group's last round is asked)
                launch_dec_sync(s, d_jobs, nj, sub_total, r, d_start, d_end, d_bytes, d_flags, d_changed);
            }
            HIP_TRY(hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            settled = !changed;
        }
        launch_dec_offsets(s, d_jobs, nj, sub_total, d_start, d_end, d_bytes, d_flags, d_recs, d_block_off, d_status, d_eob);
        launch_dec_emit(s, d_jobs, nj, sub_total, d_start, d_bytes, d_eob, d_block_off, d_status);
    }
    launch_dec_finish(s, d_jobs, nj, std::max(max_rows, 1u), std::max(max_bpl, 1u), d_status);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> status(nj);
    HIP_TRY(hipMemcpyAsync(status.data(), d_status, nj * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    static const bool trace = getenv("FPNG_AMD_TRACE") != nullptr;
    for (uint32_t k = 0; k < nj; k++) {
        if (trace)
            fprintf(stderr, "[decode] file %u: %ux%ux%u mode %u, %u subsequences, first bit %llu, device status 0x%x\n", job_file[k], jobs[k].w, jobs[k].h,
                    jobs[k].src_c, jobs[k].mode, jobs[k].n_sub, (unsigned long long)jobs[k].first_bit, status[k]);
        if (jobs[k].mode) continue;
        int32_t &st = results[job_file[k]].status;
        if (status[k] & kDecNotConverged) // (nothing else is known then: "invalid" may be a speculative decode's)
            st = FPNG_AMD_DECODE_UNDECIDED;
        else if (status[k] & kDecBadStream)
            st = fpng::FPNG_DECODE_NOT_FPNG;
        else if (!(status[k] & kDecSawEob))
            st = fpng::FPNG_DECODE_NOT_FPNG; // the stream never ended
    }
    return FPNG_AMD_OK;
}
