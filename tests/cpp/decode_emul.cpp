// decode_emul.cpp -- the GPU decoder's kernels (fpng_amd/csrc/decode.hip) run on the CPU, thread by thread, over the SAME per-thread
// code (fpng_amd/csrc/decode_core.h) and the same host-side preparation (fpng_amd_decode_plan): there is no GPU in the dev
// container, so this is where the walkers, the hand-over rules, the in-workgroup and cross-border correction rounds, the offsets
// and the emit logic are held against the CPU decoder / the reference before a GPU sees them
// (tests/test_decode_model.py).  Workgroup size and lead-in are PARAMETERS here (the kernels fix them at compile time):
// small values put many workgroup borders into small test images.  TEST INFRASTRUCTURE -- not part of the product.
#include "decode_core.h"
#include "fpng_amd.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

using namespace fpng_amd::dec;

namespace {

constexpr uint32_t kSubBits = 512;

struct HostBits { // positions relative to dword `d0` of the (zero padded) stream
    const uint32_t *dw;
    uint32_t window(uint32_t pos) const { return funnel(dw[(pos >> 5) + 1], dw[pos >> 5], pos & 31u); }
};
// a subsequence's token records (decode.hip: TokOut): an entry of two per step of the walk; at most kRecCap are kept, all are counted
struct HostRec {
    std::vector<uint64_t> *v;
    uint32_t k = 0;
    void put2(uint32_t a, uint32_t b)
    {
        if (!(a | b)) return;
        if (k < kRecCap) (*v)[k] = (uint64_t)b << 32 | a;
        k++;
    }
    uint32_t count() const { return k; }
};
// a window of the filtered stream in a tile (decode.hip: TileOut): eight bytes of slack on either side; every store of a walk must
// stay inside the walk's own territory -- the bytes of its subsequence -- or outside the window
struct HostOut {
    uint8_t *win0;              // window byte 0
    std::vector<bool> *marked;  // the window's pixels that a long match covers
    uint32_t *epx;
    bool *has_epx, *fault;
    int32_t wlen, own_lo, own_hi; // the walk's territory in window positions
    void put64(int32_t pos, uint64_t v)
    {
        if (pos < -8 || pos > wlen) *fault = true;
        for (int q = 0; q < 8; q++) {
            const int32_t x = pos + q;
            if (x >= 0 && x < wlen && (x < own_lo || x >= own_hi)) *fault = true; // another walk's byte
            win0[x] = (uint8_t)(v >> (8 * q));
        }
    }
    void entry_px(uint32_t th) { *epx = th, *has_epx = true; }
    void mark(uint32_t lo, uint32_t hi)
    {
        if (hi > marked->size()) *fault = true;
        for (uint32_t p = lo; p < hi && p < marked->size(); p++) (*marked)[p] = true;
    }
};

} // namespace

// status: the decoder's (0, FPNG_DECODE_* or FPNG_AMD_DECODE_UNDECIDED); stats[0] = correction rounds inside workgroups (max),
// stats[1] = border rounds, stats[2] = subsequences, stats[3] = subsequences corrected at least once
extern "C" int fpng_emul_decode(const uint8_t *png, uint32_t size, uint32_t desired, uint8_t *out, size_t out_cap, uint32_t *w_, uint32_t *h_, uint32_t *c_, uint32_t sub_block,
                                uint32_t lead_in, uint32_t /*unused*/, uint32_t max_border_rounds, uint32_t *stats)
{
    fpng_amd_decode_result res;
    uint32_t mode = 0, idat_ofs = 0, idat_len = 0;
    uint64_t first_bit = 0, end_limit = 0;
    std::vector<uint32_t> lut(FPNG_AMD_DECODE_LUT_WORDS);
    if (fpng_amd_decode_plan(png, size, &res, &mode, &idat_ofs, &idat_len, &first_bit, &end_limit, lut.data())) return -1000;
    *w_ = res.w, *h_ = res.h, *c_ = res.channels_in_file;
    if (res.status) return res.status;
    const uint32_t w = res.w, h = res.h, c = res.channels_in_file, bpl = w * c, stride = bpl + 1;
    const uint64_t total = (uint64_t)stride * h;
    if ((uint64_t)w * h * desired > out_cap) return -1001;
    const uint8_t *zsrc = png + idat_ofs + 8;
    std::vector<uint8_t> filt(total + 32);
    if (mode == 1) { // stored blocks of 65535 bytes, filter 0: the pixels themselves (dec_stored_kernel)
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++)
                for (uint32_t ch = 0; ch < desired; ch++) {
                    const uint64_t s = (uint64_t)y * stride + 1 + (uint64_t)x * c + ch;
                    out[((size_t)y * w + x) * desired + ch] = ch < c ? zsrc[2 + 5 * (s / 65535 + 1) + s] : 0xFF;
                }
        return 0;
    } else {
        // the stream as the kernels see it: dwords from an aligned start; shift 1 puts the first byte into the middle of a dword
        const uint32_t z_shift = 1;
        std::vector<uint32_t> zdw((idat_len + z_shift + 16) / 4 + 64, 0);
        memcpy((uint8_t *)zdw.data() + z_shift, zsrc, idat_len);
        const uint64_t z_bytes = (uint64_t)idat_len + z_shift;
        first_bit += 8 * z_shift, end_limit += 8 * z_shift;
        const uint8_t *lenof = (const uint8_t *)(lut.data() + kLutEntries);
        const uint32_t n_sub = (uint32_t)((end_limit - first_bit + kSubBits - 1) / kSubBits), nb = (n_sub + sub_block - 1) / sub_block;
        std::vector<uint32_t> info(n_sub), bytes(n_sub), eob_rel(n_sub, 0);
        std::vector<std::vector<uint64_t>> tok(n_sub, std::vector<uint64_t>(kRecCap, 0)); // the entries the settling decode of every subsequence leaves
        struct Rec {
            uint32_t sum, first_eob, first_invalid, first_overflow, entry_rel, exit_rel, want_rel;
            PhaseMap bmap;
            uint32_t left;
        };
        std::vector<Rec> recs(nb);
        uint32_t max_inner = 0, fixed_subs = 0;
        const uint32_t kUnknown = 0xFFFFFFFFu;
        // ---- dec_sync_kernel ----
        auto sync_block = [&](uint32_t blk, uint32_t round) -> bool { // returns "changed"
            const uint32_t local0 = blk * sub_block;
            uint32_t want0 = 0;
            bool cand_first = false;
            if (round) {
                // (the file's first subsequence starts at the stream's first token: no border in front of the first block)
                const uint32_t prev = !local0 ? recs[blk].entry_rel : (recs[blk].want_rel != kUnknown ? recs[blk].want_rel : recs[blk - 1].exit_rel); // (dec_chain_kernel's word, else the neighbour's)
                if (recs[blk].entry_rel == prev && pm_count(recs[blk].bmap)) return false;
                want0 = prev;
                cand_first = pm_count(recs[blk].bmap) > 1 || (!pm_count(recs[blk].bmap) && recs[blk].left == 2); // its corrections did not end in round 0, or it needed its phase maps before
            }
            const uint32_t lead0 = local0 ? lead_in : 0u;
            const uint64_t first_nominal = first_bit + (uint64_t)local0 * kSubBits, d0 = (first_nominal - lead0) >> 5, base = d0 << 5;
            HostBits in = {zdw.data() + d0};
            const uint64_t lim64 = end_limit - base;
            const uint32_t data_limit = lim64 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)lim64;
            const uint32_t nthreads = std::min(sub_block, n_sub - local0);
            std::vector<SubState> st(nthreads);
            std::vector<uint32_t> s_end(nthreads);
            std::vector<bool> dirty(nthreads, false), ever(nthreads, false);
            auto nominal_of = [&](uint32_t t) { return (uint32_t)(first_nominal - base) + t * kSubBits; };
            for (uint32_t t = 0; t < nthreads; t++) {
                const uint32_t i = local0 + t, nominal = nominal_of(t), boundary = nominal + kSubBits;
                if (!round) {
                    HostRec rec = {&tok[i]};
                    // (the kernel chooses the walk's form by what the lead-ins of a wave met: here by the thread's own, and by its number
                    //  where it met nothing -- both forms must settle on the same tokens)
                    uint32_t p0 = nominal, gen = 0;
                    if (i) p0 = sub_lead<VoteAlone>(in, lut.data(), lenof, nominal - lead_in, nominal, data_limit, gen);
                    sub_main<VoteAlone>(in, lut.data(), lenof, p0, boundary, data_limit, st[t], rec, gen != 0 || (i & 1));
                    dirty[t] = true;
                } else {
                    const uint32_t v = info[i];
                    st[t].start = nominal + info_start(v), st[t].end = boundary + info_end(v), st[t].nrec = info_nrec(v);
                    st[t].c.bytes = bytes[i], st[t].c.flags = info_flags(v);
                    st[t].c.eob = (st[t].c.flags & kSubEob) ? nominal + eob_rel[i] : 0u;
                }
                s_end[t] = st[t].end;
            }
            want0 += nominal_of(0);
            uint32_t inner = 0;
            PhaseMap bmap = pm_none();
            bool cand_done = false;
            uint32_t unsettled = 0; // 1: many threads to correct, 2: corrections that do not end (decode.hip: kLeftMany, kLeftCrawling)
            for (uint32_t it = 0;; it++) {
                std::vector<uint32_t> want(nthreads);
                bool any = false;
                for (uint32_t t = 0; t < nthreads; t++) { // (all threads look before any of them writes: the kernel's barrier)
                    want[t] = t ? s_end[t - 1] : (round ? want0 : st[t].start);
                    any |= want[t] != st[t].start;
                }
                if (!any) break;
                inner++;
                uint32_t n_need = 0;
                for (uint32_t t = 0; t < nthreads; t++) n_need += want[t] != st[t].start;
                if (!round && it >= kRefixRounds) { // round 0's kernel is kept lean: the block is left to round 1
                    unsettled = 2;
                    break;
                }
                const bool cand_now = round && !cand_done && (it >= kRefixRounds || (cand_first && it >= 1)); // (thread 0 takes its wanted start in step 0)
                if (!cand_now) {
                    // (the kernel gathers them into one wave from four on, else decodes them again where they are)
                    (void)n_need;
                    for (uint32_t t = 0; t < nthreads; t++)
                        if (want[t] != st[t].start) {
                            HostRec rec = {&tok[local0 + t]};
                            sub_redo<VoteAlone>(in, lut.data(), lenof, want[t], nominal_of(t) + kSubBits, data_limit, st[t], rec, (t ^ it) & 1);
                            s_end[t] = st[t].end;
                            dirty[t] = true;
                            if (!ever[t]) ever[t] = true, fixed_subs++;
                        }
                    continue;
                }
                // ---- phase maps (decode_core.h) ----
                cand_done = true;
                std::vector<PhaseMap> map(nthreads), pub(nthreads);
                for (uint32_t t = 0; t < nthreads; t++) map[t] = pm_one(st[t].start - nominal_of(t), st[t].end - (nominal_of(t) + kSubBits));
                // every phase a decoder can arrive in at a thread's nominal bit: lead-ins from 18 consecutive bits (pm_seed)
                for (uint32_t t = 0; t < nthreads; t++)
                    if (local0 + t && lead_in >= 18) pm_seed<VoteAlone>(in, lut.data(), lenof, nominal_of(t) - lead_in, nominal_of(t), nominal_of(t) + kSubBits, data_limit, map[t]);
                for (uint32_t step = 0; step < kCandGrowSteps; step++) {
                    pub = map;
                    bool grew = false;
                    for (uint32_t t = 1; t < nthreads; t++)
                        grew |= pm_grow<VoteAlone>(in, lut.data(), lenof, pub[t - 1], nominal_of(t), nominal_of(t) + kSubBits, data_limit, map[t]);
                    inner++;
                    if (!grew) break;
                }
                // prefix composition (shuffles and wave totals in the kernel): g[t] = what the subsequences 0..t do to the phases thread 0 knows
                std::vector<PhaseMap> g = map;
                for (uint32_t t = 1; t < nthreads; t++) g[t] = pm_compose(g[t - 1], map[t]);
                bmap = nthreads == sub_block ? g[nthreads - 1] : pm_none();
                const uint32_t start0 = st[0].start - nominal_of(0);
                for (uint32_t t = 1; t < nthreads; t++) {
                    const uint32_t srel = pm_at(g[t - 1], start0);
                    if (srel == kPhaseUnknown) continue;
                    const uint32_t ws = nominal_of(t) + srel;
                    if (ws == st[t].start) continue;
                    HostRec rec = {&tok[local0 + t]};
                    sub_redo<VoteAlone>(in, lut.data(), lenof, ws, nominal_of(t) + kSubBits, data_limit, st[t], rec, true);
                    dirty[t] = true;
                    if (!ever[t]) ever[t] = true, fixed_subs++;
                }
                for (uint32_t t = 0; t < nthreads; t++) s_end[t] = st[t].end;
            }
            max_inner = std::max(max_inner, inner);
            bool any_dirty = false;
            for (uint32_t t = 0; t < nthreads; t++)
                if (dirty[t]) {
                    const uint32_t i = local0 + t, nominal = nominal_of(t);
                    info[i] = pack_info(st[t].start - nominal, st[t].end - (nominal + kSubBits), st[t].c, st[t].nrec);
                    if (info_start(info[i]) != st[t].start - nominal || info_end(info[i]) != st[t].end - (nominal + kSubBits) || info_nrec(info[i]) != st[t].nrec ||
                        info_flags(info[i]) != st[t].c.flags)
                        throw 1; // a field of the record overflowed
                    bytes[i] = st[t].c.bytes;
                    if (st[t].c.flags & kSubEob) eob_rel[i] = st[t].c.eob - nominal;
                    any_dirty = true;
                }
            if (!any_dirty) return false;
            Rec r = {0, sub_block, sub_block, sub_block, st[0].start - nominal_of(0), 0, kUnknown, pm_none(), unsettled};
            for (uint32_t t = 0; t < nthreads; t++) {
                r.sum += st[t].c.bytes;
                if ((st[t].c.flags & kSubEob) && r.first_eob == sub_block) r.first_eob = t;
                if ((st[t].c.flags & kSubInvalid) && r.first_invalid == sub_block) r.first_invalid = t;
                if ((st[t].c.flags & kSubOverflow) && r.first_overflow == sub_block) r.first_overflow = t;
            }
            r.exit_rel = nthreads == sub_block ? s_end[sub_block - 1] - (nominal_of(0) + sub_block * kSubBits) : 0u;
            r.bmap = (pm_count(bmap) || unsettled) ? bmap : pm_one(r.entry_rel, r.exit_rel);
            recs[blk] = r;
            return round != 0;
        };
        try {
            for (uint32_t b = 0; b < nb; b++) sync_block(b, 0);
        } catch (int) {
            return -1002;
        }
        // (from here on `throw 3`: dec_subscan_kernel's windows)
        // ---- dec_chain_kernel: every block's true entry as far as the blocks' maps tell it (a block whose map does not know the phase
        //      it is entered in hands on its present exit: the neighbour's word, as before) ----
        auto block_fn = [&](uint32_t b, uint32_t x) {
            const uint32_t e = pm_at(recs[b].bmap, x);
            return e != kPhaseUnknown ? e : (recs[b].exit_rel & 31u);
        };
        auto chain = [&]() {
            uint32_t cur = recs[0].entry_rel;
            for (uint32_t b = 0; b < nb; b++) {
                recs[b].want_rel = cur;
                cur = block_fn(b, cur);
            }
            // the kernel's PARALLEL form of the same walk (chunks of blocks per thread, the chunks' functions composed by a
            // Hillis-Steele scan over kThreads threads) must ask for the same entries
            const uint32_t kThreads = 256, per = (nb + kThreads - 1) / kThreads;
            std::vector<PhaseMap> f(kThreads), g(kThreads);
            for (uint32_t t = 0; t < kThreads; t++) {
                f[t] = pm_none();
                for (uint32_t x = 0; x < kPhases; x++) pm_set(f[t], x, x);
                for (uint32_t b = std::min(t * per, nb); b < std::min(t * per + per, nb); b++) {
                    PhaseMap n2 = f[t];
                    for (uint32_t x = 0; x < kPhases; x++) pm_set(n2, x, block_fn(b, pm_at(f[t], x)));
                    f[t] = n2;
                }
            }
            for (uint32_t d = 1; d < kThreads; d <<= 1) {
                g = f;
                for (uint32_t t = d; t < kThreads; t++) f[t] = pm_compose(g[t - d], g[t]);
            }
            for (uint32_t t = 0; t < kThreads; t++) {
                uint32_t ph = recs[0].entry_rel;
                if (t) ph = pm_at(f[t - 1], ph);
                for (uint32_t b = std::min(t * per, nb); b < std::min(t * per + per, nb); b++) {
                    if (recs[b].want_rel != ph) throw 2;
                    ph = block_fn(b, ph);
                }
            }
        };
        uint32_t border_rounds = 0;
        for (uint32_t r = 1; r <= max_border_rounds; r++) {
            try {
                if (r >= 2) chain(); // (the blocks' maps have more than one pair only behind round 1)
            } catch (int) {
                return -1006; // the parallel form of the chain disagrees with the walk
            }
            // (the workgroups of one launch run at the same time; last to first, each one sees its predecessor's record of the
            //  launch before -- the least favourable of the schedules the kernel allows)
            bool changed = false;
            try {
                for (uint32_t b = nb; b-- > 0;) changed |= sync_block(b, r);
            } catch (int) {
                return -1002;
            }
            if (!changed) break;
            border_rounds++;
        }
        if (stats) stats[0] = max_inner, stats[1] = border_rounds, stats[2] = n_sub, stats[3] = fixed_subs;
        // ---- dec_offsets_kernel ----
        uint32_t status = 0, last_blk = nb, last_local = 0;
        for (uint32_t b = 0; b < nb && last_blk == nb; b++)
            if (recs[b].first_eob < sub_block) last_blk = b, last_local = recs[b].first_eob;
        std::vector<uint64_t> block_off(nb, 0);
        uint64_t acc = 0;
        for (uint32_t b = 0; b < nb && b <= last_blk; b++) {
            if ((b && recs[b].entry_rel != recs[b - 1].exit_rel) || !pm_count(recs[b].bmap)) status |= 1; // not converged
            block_off[b] = acc;
            if (b < last_blk) {
                acc += recs[b].sum;
                if (recs[b].first_invalid < sub_block) status |= 2;
                if (recs[b].first_overflow < sub_block) status |= 16;
            } else {
                if (recs[b].first_invalid <= last_local) status |= 2;
                if (recs[b].first_overflow <= last_local) status |= 16;
            }
        }
        if (last_blk < nb)
            for (uint32_t k = 0; k <= last_local; k++) acc += bytes[last_blk * sub_block + k];
        else
            status |= 2;
        if (acc != total) status |= 2;
        const uint32_t eob_index = last_blk * sub_block + last_local;
        // the stream must end 4 bytes (the Adler-32) before the IDAT does (eob_status() of decode.hip)
        if (last_blk < nb && ((first_bit + (uint64_t)eob_index * kSubBits + eob_rel[eob_index] + 7) >> 3) + 4 != z_bytes) status |= 2;
        if (status & (1 | 16)) return FPNG_AMD_DECODE_UNDECIDED; // (decode_api.cpp: not converged / left to the CPU decoder go first)
        if (status) return 1; // FPNG_DECODE_NOT_FPNG
        // ---- dec_subscan_kernel: offsets, the literal bytes in front, and the windows (decode_core.h) that begin in each subsequence's output ----
        // (the kernel's tiles: column blocks of 256 pixels)
        const uint32_t cbw = 256u * c, ncb = (w + 255u) / 256u;
        std::vector<uint32_t> rel(n_sub, 0), lastpx(n_sub, 0), win((size_t)h * ncb, 0xFFFFFFFFu);
        struct ResumePoint {
            uint32_t k, crel, th;
        };
        std::vector<ResumePoint> resume((size_t)h * ncb, ResumePoint{kNoResume, 0, 0});
        try {
        for (uint32_t b = 0; b <= last_blk; b++) {
            uint32_t before = 0;
            for (uint32_t t = 0; t < sub_block; t++) {
                const uint32_t i = b * sub_block + t;
                if (i >= n_sub || i > eob_index) break;
                rel[i] = before;
                const uint32_t nent_i = info_nrec(info[i]);
                lastpx[i] = (nent_i && needs_lastpx(tok[i][0], tok[i][1], nent_i))
                                ? lookback_lastpx(
                                      i, [&](uint32_t k) { return info_nrec(info[k]); }, [&](uint32_t k, uint32_t e) { return tok[k][e]; })
                                : 0xDEADBEEFu; // (never looked at: a walk that did would write wrong pixels here)
                // (a subsequence that covers many windows leaves each of them a resume point: decode_core.h; the emulator's threshold is
                //  low -- small test images have small windows' worth of flat rows)
                const bool big = bytes[i] >= std::min<uint32_t>(kResumeMinBytes, 2 * cbw);
                ResumeWalk rw = resume_begin(lastpx[i] == 0xDEADBEEFu ? 0u : lastpx[i]);
                const uint64_t off_i = block_off[b] + before;
                for_windows_starting_in(off_i, bytes[i], cbw, ncb, stride, h, [&](uint32_t y, uint32_t cb) {
                    if (win[(size_t)y * ncb + cb] != 0xFFFFFFFFu) throw 3; // two subsequences claim one window
                    win[(size_t)y * ncb + cb] = i;
                    if (big) {
                        const uint32_t d = (uint32_t)((uint64_t)y * stride + (cb ? 1u + cb * cbw : 0u) - off_i);
                        resume_seek(rw, d, std::min(nent_i, kRecCap), [&](uint32_t k) { return tok[i][k]; });
                        resume[(size_t)y * ncb + cb] = ResumePoint{rw.sk, rw.sc - d, rw.sth};
                    }
                });
                before += bytes[i];
            }
        }
        } catch (int) {
            return -1007; // two subsequences claim one window
        }
        // ---- dec_unfilter_kernel's tiles, first half: every window filled from the records of the subsequences that reach into it -- from
        //      the one it begins in to the one the next window begins in, a walk each (decode_core.h) -- then the pixels of the long matches ----
        uint32_t err = 0;
        bool fault = false;
        std::vector<uint8_t> tile(8 + cbw + 1 + 8);
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t cb = 0; cb < ncb; cb++) {
                const Window wd = window_of(y, cb, cbw, stride);
                const size_t wi = (size_t)y * ncb + cb;
                const uint32_t i0 = win[wi];
                if (i0 == 0xFFFFFFFFu) return -1005; // (the stream covers the image: every window begins somewhere)
                uint32_t i1 = wi + 1 < (size_t)h * ncb ? win[wi + 1] : 0xFFFFFFFFu;
                i1 = std::min(i1, eob_index);
                std::fill(tile.begin(), tile.end(), (uint8_t)0xCD);
                const uint32_t fb = cb ? 0u : 1u, npx = (wd.wlen - fb) / c;
                std::vector<bool> marked(npx, false);
                uint32_t epx = 0;
                bool has_epx = false;
                int32_t covered = 0; // window bytes [0, covered) belong to walks so far
                for (uint32_t i = i0; i <= i1; i++) {
                    const uint64_t off = block_off[i / sub_block] + rel[i];
                    if (off >= wd.ws + wd.wlen) break;
                    WalkState ws;
                    ws.c = ws.c0 = (int32_t)(int64_t)(off - wd.ws), ws.tl = 0, ws.th = lastpx[i], ws.err = 0;
                    if (i == i0 ? ws.c > 0 : ws.c != covered) return -1005; // a gap between two walks
                    const int32_t own_lo = ws.c0, own_hi = ws.c0 + (int32_t)std::min<uint64_t>(bytes[i], 1u << 30);
                    uint32_t kstart = 0;
                    if (i == i0 && resume[wi].k != kNoResume) // the window's first walk begins at its resume point
                        kstart = resume[wi].k, ws.c = (int32_t)resume[wi].crel, ws.c0 = ws.c - 8, ws.th = resume[wi].th; // (c0: decode.hip, fill_tile)
                    if (i == i0 && ws.c > 0) return -1005;
                    HostOut out = {tile.data() + 8, &marked, &epx, &has_epx, &fault, (int32_t)wd.wlen, own_lo, own_hi};
                    const uint32_t nent = std::min(info_nrec(info[i]), kRecCap);
                    for (uint32_t k = kstart; k < nent; k++) { // (the kernel goes on to the end of its batch of entries: behind the window's end every store lands in the slack)
                        const uint64_t en = tok[i][k];
                        const uint32_t a = (uint32_t)en, b2 = (uint32_t)(en >> 32);
                        // (the kernel takes a straight-line step when the entries of ALL lanes of the wave allow it -- all forms must do the
                        //  same -- and drops walk_apply's hold-back behind a walk's first batch of eight entries)
                        const bool hold = k - kstart < 8, lits = !((a | b2) & kRecRun) && ((k ^ i) & 2);
                        const bool plain = (c == 4 ? entry_plain<4>(a, b2) : entry_plain<3>(a, b2)) && ((k ^ i) & 1);
                        if (lits) {
                            if (hold) walk_entry_literals<true>(a, b2, ws, wd, out); else walk_entry_literals<false>(a, b2, ws, wd, out);
                        } else if (plain) {
                            if (c == 4) {
                                if (hold) walk_entry_plain<4, true>(a, b2, ws, wd, stride, out); else walk_entry_plain<4, false>(a, b2, ws, wd, stride, out);
                            } else {
                                if (hold) walk_entry_plain<3, true>(a, b2, ws, wd, stride, out); else walk_entry_plain<3, false>(a, b2, ws, wd, stride, out);
                            }
                        } else if (entry_long_matches(a, b2) && ((k ^ i) & 4)) {
                            if (c == 4) walk_entry_long<4>(a, b2, ws, wd, stride, out); else walk_entry_long<3>(a, b2, ws, wd, stride, out);
                        } else if ((c == 4 ? entry_mixed<4>(a, b2) : entry_mixed<3>(a, b2)) && ((k ^ i) & 8)) {
                            if (c == 4) walk_entry_mixed<4>(a, b2, ws, wd, stride, out); else walk_entry_mixed<3>(a, b2, ws, wd, stride, out);
                        } else if (c == 4)
                            walk_entry<4>(en, ws, wd, stride, out);
                        else
                            walk_entry<3>(en, ws, wd, stride, out);
                    }
                    if (ws.c != own_hi) return -1008; // the records do not add up to the subsequence's byte count
                    covered = ws.c;
                    err |= ws.err;
                }
                if (covered < (int32_t)wd.wlen) return -1005; // a byte of the window nobody wrote
                // the long matches' pixels (decode.hip: propagate_matches): the nearest unmarked pixel to the left, else the entry pixel
                uint8_t *data = tile.data() + 8 + fb;
                for (uint32_t p = 0; p < npx; p++) {
                    if (!marked[p]) continue;
                    if (p == 0 && !has_epx && !(err & (2u | kEmitLeaveToCpu))) return -1009; // a window that begins inside a match nobody handed over
                    const uint32_t px = c == 4 ? epx : epx >> 8;
                    for (uint32_t q = 0; q < c; q++) data[(size_t)p * c + q] = p ? data[(size_t)(p - 1) * c + q] : (uint8_t)(px >> (8 * q));
                }
                memcpy(filt.data() + wd.ws, tile.data() + 8, wd.wlen);
            }
        if (fault) return -1004;
        if (err & kEmitLeaveToCpu) return FPNG_AMD_DECODE_UNDECIDED; // (decode_api.cpp: kDecStalled goes first -- a match at a row's first pixel is the CPU decoder's)
        if (err & 2) return 1;
    }
    // ---- Up filter undone, channel conversion (dec_unfilter_kernel) ----
    std::vector<uint8_t> prev(bpl, 0), cur(bpl);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *f = filt.data() + (size_t)y * stride + 1;
        if (f[-1] != (y ? 2 : 0)) return 1; // the row's filter literal: 0, then 2 = Up
        for (uint32_t j = 0; j < bpl; j++) cur[j] = (uint8_t)(prev[j] + f[j]);
        uint8_t *o = out + (size_t)y * w * desired;
        for (uint32_t x = 0; x < w; x++)
            for (uint32_t ch = 0; ch < desired; ch++) o[(size_t)x * desired + ch] = ch < c ? cur[(size_t)x * c + ch] : 0xFF;
        prev.swap(cur);
    }
    return 0;
}

// the phase map helpers of decode_core.h (tests/test_decode_model.py): set / at round trips, composition against its definition and
// its associativity, on `n` random maps drawn from `seed`; 0 = all held
extern "C" int fpng_emul_phase_map_selftest(uint32_t seed, uint32_t n)
{
    uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() {
        st ^= st << 13, st ^= st >> 7, st ^= st << 17;
        return (uint32_t)(st >> 32);
    };
    auto random_map = [&]() {
        PhaseMap m = pm_none();
        for (uint32_t x = 0; x < kPhases; x++)
            if (rnd() % 3) pm_set(m, x, rnd() % kPhases);
        return m;
    };
    for (uint32_t k = 0; k < n; k++) {
        uint8_t ref[kPhases];
        PhaseMap m = pm_none();
        if (pm_count(m)) return 1;
        uint32_t known = 0;
        for (uint32_t x = 0; x < kPhases; x++) ref[x] = kPhaseUnknown;
        for (uint32_t j = 0; j < 40; j++) {
            const uint32_t x = rnd() % kPhases, v = rnd() % 4 ? rnd() % kPhases : kPhaseUnknown;
            pm_set(m, x, v), ref[x] = (uint8_t)v;
        }
        for (uint32_t x = 0; x < kPhases; x++) {
            if (pm_at(m, x) != ref[x]) return 2;
            known += ref[x] != kPhaseUnknown;
        }
        if (pm_count(m) != known) return 3;
        const PhaseMap a = random_map(), b = random_map(), c = random_map();
        const PhaseMap ab = pm_compose(a, b), bc = pm_compose(b, c), l = pm_compose(ab, c), r = pm_compose(a, bc);
        for (uint32_t x = 0; x < kPhases; x++) {
            const uint32_t ax = pm_at(a, x), want = ax == kPhaseUnknown ? kPhaseUnknown : pm_at(b, ax);
            if (pm_at(ab, x) != want) return 4;
            if (pm_at(l, x) != pm_at(r, x)) return 5;
        }
        const uint32_t s0 = rnd() % kPhases, e0 = rnd() % kPhases;
        const PhaseMap one = pm_one(s0, e0);
        if (pm_count(one) != 1 || pm_at(one, s0) != e0) return 6;
    }
    return 0;
}


// The tokens of a dynamic-block fpng file, one by one (tests/token_mutator.py edits large files with it): kind 0 = literal (value =
// the byte), 1 = match (value = its length, aux = the distance bit), 2 = end of block; bitpos = the token's first bit, counted from
// the zlib stream's first byte.  Returns the number of tokens (the end-of-block symbol included), -1 if the file has no such
// stream, -2 if it derails, -3 if `cap` is too small.
extern "C" long long fpng_emul_tokenize(const uint8_t *png, uint32_t size, long long cap, uint8_t *kind, uint16_t *value, uint8_t *aux, uint64_t *bitpos)
{
    fpng_amd_decode_result res;
    uint32_t mode = 0, idat_ofs = 0, idat_len = 0;
    uint64_t first_bit = 0, end_limit = 0;
    std::vector<uint32_t> lut(FPNG_AMD_DECODE_LUT_WORDS);
    if (fpng_amd_decode_plan(png, size, &res, &mode, &idat_ofs, &idat_len, &first_bit, &end_limit, lut.data()) || res.status || mode) return -1;
    std::vector<uint32_t> zdw((size_t)idat_len / 4 + 8, 0);
    memcpy(zdw.data(), png + idat_ofs + 8, idat_len);
    const uint8_t *lenof = (const uint8_t *)(lut.data() + kLutEntries);
    long long n = 0;
    uint64_t pos = first_bit;
    for (;;) {
        if (pos >= end_limit) return -2;
        if (n >= cap) return -3;
        const uint64_t d = pos >> 5;
        const uint32_t w = funnel(zdw[d + 1], zdw[d], (uint32_t)(pos & 31));
        // (decode_core.h: a simple token's entry holds ALL the bits it takes in its upper four, any other its code bits in 15..12)
        const uint32_t e = lut[w & (kLutEntries - 1)], adv = e >> 28, nl = (e >> 26) & 3u;
        if (!e) return -2;
        bitpos[n] = pos;
        if (nl) {
            const uint32_t b = e & 255u;
            kind[n] = 0, value[n] = (uint16_t)b, aux[n] = 0;
            pos += lenof[b];
        } else if (e & kEntMatch) {
            const uint32_t L = adv ? adv - 1u : (e >> 12) & 15u, xb = (e >> 9) & 7u;
            kind[n] = 1, value[n] = (uint16_t)((e & 511u) + ((w >> L) & ((1u << xb) - 1u))), aux[n] = (uint8_t)((w >> (L + xb)) & 1u);
            pos += L + xb + 1;
        } else {
            kind[n] = 2, value[n] = 0, aux[n] = 0;
            return n + 1;
        }
        n++;
    }
}
