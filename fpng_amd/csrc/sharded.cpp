// sharded.cpp -- ONE image over several GPUs behind the C ABI (the Python mirror is fpng_amd/sharded.py).
//
// Every rank holds a band of rows and an encoder; the stream stays ONE IDAT / one zlib stream / one Deflate block (reference
// src/fpng.cpp:1764-1800; its decoder rejects a second IDAT, :3032-3033), so bands meet at bit granularity:
//   0. (2-pass) histogram of the band, all-reduce of 288 counters: every rank builds the same table
//   1. encode the band into local streams; its counts come back to the host
//   2. all-gather of one record per rank -> every rank derives every band's start bit, the image's Adler-32 and the
//      reference's compressed-or-stored decision (fpng_amd_plan_bands)
//   3. place the band in a window of whole 16-byte pieces of the file (CRC ranges hung off the window's own end: one raw CRC
//      per band), all-gather of {raw CRC, window end}
//   4. windows to the root, straight into their place in the file; a window's first piece travels on its own when the band
//      shares it with its predecessor and is OR-ed in afterwards
//   5. the root writes the 58-byte head and the 20-byte tail (Adler-32, IDAT CRC from the bands' values, IEND).
// Exchanges go through an fpng_amd_transport; the RCCL one (ncclAllGather / ncclAllReduce / ncclSend / ncclRecv on the
// encoder's stream) is loaded with dlopen so that the library does not depend on librccl.
#include "encoder.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

using namespace fpng_amd;

namespace {

struct Record { // what the ranks exchange in step 2 (64 bytes)
    uint64_t token_bits, adler_len;
    uint32_t adler_s1, adler_s2, last_unit_bits, first_token_bit, eob_bits, y0, y1, pad[5];
};
static_assert(sizeof(Record) == 64, "record layout");
struct CrcRecord { // step 3
    uint64_t end_offset;
    uint32_t raw_crc, pad;
};

#define T_TRY(expr)                                                                         \
    do {                                                                                    \
        int rc_ = (expr);                                                                   \
        if (rc_) return rc_ < 0 ? rc_ : fail(FPNG_AMD_ERR_HIP, "transport call failed: " #expr); \
    } while (0)

} // namespace

extern "C" int fpng_amd_encode_image_sharded(fpng_amd_encoder *e, const fpng_amd_transport *t, const fpng_amd_band *band, uint32_t flags,
                                             int root, uint8_t *d_png, size_t png_cap, size_t *png_size)
{
    if (!e || !t || !band || !t->all_gather || !t->send || !t->recv || !t->group_begin || !t->group_end)
        return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument / incomplete transport");
    const int rank = t->rank, world = t->world;
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return fail(FPNG_AMD_ERR_INVALID_ARG, "bad rank / world / root");
    if (flags & ~FPNG_AMD_ENCODE_SLOWER) return fail(FPNG_AMD_ERR_INVALID_ARG, "flags: 0 or FPNG_AMD_ENCODE_SLOWER");
    const uint32_t w = band->w, h = band->h_total, c = band->num_chans;
    int rc = check_dims(w, h, c);
    if (rc) return rc;
    if (band->y1 < band->y0 || band->y1 > h) return fail(FPNG_AMD_ERR_INVALID_ARG, "out-of-range band");
    const bool two_pass = (flags & FPNG_AMD_ENCODE_SLOWER) != 0, have_rows = band->y1 > band->y0, is_root = rank == root;
    if (is_root && (!d_png || !png_size || png_cap < fpng_amd_max_encoded_size(w, h, c) + 64 || ((uintptr_t)d_png & 15)))
        return fail(FPNG_AMD_ERR_INVALID_ARG, "root: d_png (16-byte aligned) / png_cap >= fpng_amd_max_encoded_size() + 64 / png_size");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const size_t bpl = (size_t)w * c;
    fpng_amd_sharded_report &rep = e->sharded_report;
    std::memset(&rep, 0, sizeof rep);
    // FPNG_AMD_TRACE=1: where a call's time goes, on stderr (microseconds from its start)
    static const bool trace = getenv("FPNG_AMD_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[sharded rank %d] %-28s %8.0f us\n", rank, what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count());
    };

    // exchange scratch: [hist 288 u32][my record][all records][my crc record][all crc records][heads 16 B x world]
    const size_t o_hist = 0, o_rec = o_hist + 288 * 4, o_recs = o_rec + sizeof(Record), o_crc = o_recs + sizeof(Record) * world,
                 o_crcs = o_crc + sizeof(CrcRecord), o_heads = o_crcs + sizeof(CrcRecord) * world, total = o_heads + 16 * (size_t)world + 128;
    if ((rc = e->d_xchg.ensure(total)) || (rc = e->h_xchg.ensure(total))) return rc;
    uint8_t *dx = e->d_xchg.p, *hx = e->h_xchg.p;

    // ---- 0/1: histogram all-reduce (2-pass), encode the band ----
    uint32_t *d_hist = nullptr;
    if (two_pass) {
        if (!t->all_reduce_sum_u32) return fail(FPNG_AMD_ERR_INVALID_ARG, "2-pass needs transport.all_reduce_sum_u32");
        d_hist = (uint32_t *)(dx + o_hist);
        if (have_rows) {
            if ((rc = fpng_amd_band_hist(e, band, d_hist))) return rc;
        } else
            HIP_TRY(hipMemsetAsync(d_hist, 0, 288 * 4, s));
        T_TRY(t->all_reduce_sum_u32(t->ctx, d_hist, 288, s));
        rep.collectives++;
    }
    fpng_amd_band_stats st;
    std::memset(&st, 0, sizeof st);
    lap("histogram all-reduce");
    if (have_rows && (rc = fpng_amd_band_encode(e, band, flags, d_hist, &st))) return rc;
    lap("band encoded, counts here");

    // ---- 2: records ----
    Record *mine = (Record *)(hx + o_rec);
    std::memset(mine, 0, sizeof *mine);
    mine->token_bits = st.token_bits, mine->adler_len = st.adler_len, mine->adler_s1 = st.adler_s1, mine->adler_s2 = st.adler_s2;
    mine->last_unit_bits = st.last_unit_bits, mine->first_token_bit = st.first_token_bit, mine->eob_bits = st.eob_bits;
    mine->y0 = band->y0, mine->y1 = band->y1;
    HIP_TRY(hipMemcpyAsync(dx + o_rec, mine, sizeof(Record), hipMemcpyHostToDevice, s));
    T_TRY(t->all_gather(t->ctx, dx + o_rec, dx + o_recs, sizeof(Record), s));
    rep.collectives++;
    HIP_TRY(hipMemcpyAsync(hx + o_recs, dx + o_recs, sizeof(Record) * world, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    lap("records gathered");
    const Record *recs = (const Record *)(hx + o_recs);
    std::vector<int> order(world); // ranks in row order
    for (int r = 0; r < world; r++) order[r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return recs[a].y0 != recs[b].y0 ? recs[a].y0 < recs[b].y0 : recs[a].y1 < recs[b].y1; });
    std::vector<fpng_amd_band_stats> stats(world);
    uint32_t covered = 0;
    for (int k = 0; k < world; k++) {
        const Record &q = recs[order[k]];
        if (q.y1 > q.y0 && q.y0 != covered) return fail(FPNG_AMD_ERR_INVALID_ARG, "the ranks' bands do not tile the image");
        if (q.y1 > q.y0) covered = q.y1;
        fpng_amd_band_stats &b = stats[k];
        std::memset(&b, 0, sizeof b);
        b.token_bits = q.token_bits, b.adler_len = q.adler_len, b.adler_s1 = q.adler_s1, b.adler_s2 = q.adler_s2;
        b.last_unit_bits = q.last_unit_bits, b.first_token_bit = q.first_token_bit, b.eob_bits = q.eob_bits;
    }
    if (covered != h) return fail(FPNG_AMD_ERR_INVALID_ARG, "the ranks' bands do not cover the image");
    std::vector<uint64_t> start_bits(world);
    fpng_amd_band_plan plan;
    if ((rc = fpng_amd_plan_bands(stats.data(), (uint32_t)world, w, h, c, flags, start_bits.data(), &plan))) return rc;
    int my_pos = 0, first_pos = -1;
    for (int k = 0; k < world; k++) {
        if (order[k] == rank) my_pos = k;
        if (first_pos < 0 && stats[k].adler_len) first_pos = k;
    }
    const uint32_t eob_bits = stats[first_pos].eob_bits;

    rep.stored = plan.stored ? 1u : 0u;
    if (plan.stored) {
        // ---- the reference's stored-block outcome (incompressible image): no bit seams; the rows go to the root, which
        //      encodes the image whole (the stored kernels) ----
        if (is_root) {
            if ((rc = e->d_stage_in.ensure(bpl * h + 16))) return rc;
            T_TRY(t->group_begin(t->ctx));
            for (int r = 0; r < world; r++) {
                const Record &q = recs[r];
                if (q.y1 <= q.y0) continue;
                uint8_t *dst = e->d_stage_in.p + (size_t)q.y0 * bpl;
                rep.root_staged_bytes += (size_t)(q.y1 - q.y0) * bpl; // (the rows meet in a staging buffer: the stored kernels then write the file)
                if (r == rank)
                    HIP_TRY(hipMemcpyAsync(dst, band->d_rows, (size_t)(q.y1 - q.y0) * bpl, hipMemcpyDeviceToDevice, s));
                else
                    T_TRY(t->recv(t->ctx, dst, (size_t)(q.y1 - q.y0) * bpl, r, s));
            }
            T_TRY(t->group_end(t->ctx));
            fpng_amd_image im;
            im.d_pixels = e->d_stage_in.p, im.w = w, im.h = h, im.num_chans = c, im.d_out = d_png, im.out_cap = png_cap;
            fpng_amd_result res;
            if ((rc = fpng_amd_encode_batch_async(e, &im, 1, FPNG_AMD_FORCE_UNCOMPRESSED)) || (rc = fpng_amd_encode_finish(e, &res, 1))) return rc;
            if (res.status) return fail(FPNG_AMD_ERR_HIP, "device reported an encode failure");
            *png_size = (size_t)res.png_size;
        } else {
            T_TRY(t->group_begin(t->ctx));
            if (have_rows) T_TRY(t->send(t->ctx, band->d_rows, (size_t)(band->y1 - band->y0) * bpl, root, s));
            if (have_rows) rep.sent_bytes += (size_t)(band->y1 - band->y0) * bpl;
            T_TRY(t->group_end(t->ctx));
            HIP_TRY(hipStreamSynchronize(s));
        }
        return FPNG_AMD_OK;
    }

    // ---- 3: placement; the window geometry of every band follows from the plan ----
    struct Geo {
        uint64_t off = 0;
        size_t bytes = 0;
        uint32_t head = 0;
    };
    std::vector<Geo> geo(world); // by position
    for (int k = 0; k < world; k++)
        if (stats[k].adler_len)
            fpng_amd_band_window(k == first_pos, recs[order[k]].y1 == h, start_bits[k], stats[k].token_bits, eob_bits, &geo[k].off, &geo[k].bytes, &geo[k].head);
    CrcRecord *my_crc = (CrcRecord *)(hx + o_crc);
    std::memset(my_crc, 0, sizeof *my_crc);
    uint8_t *d_window = nullptr;
    if (have_rows) {
        const Geo &g = geo[my_pos];
        if (is_root) {
            d_window = d_png + g.off; // the root's own band goes straight into the file
        } else {
            if ((rc = e->d_stage_out.ensure(g.bytes + 64))) return rc;
            d_window = e->d_stage_out.p;
        }
        uint64_t off = 0;
        size_t nbytes = 0;
        if ((rc = fpng_amd_band_place(e, band, start_bits[my_pos], 0, d_window, is_root ? png_cap - g.off : e->d_stage_out.cap, &off, &nbytes))) return rc;
        if (off != g.off || nbytes != g.bytes) return fail(FPNG_AMD_ERR_HIP, "band window differs from the plan");
        rep.own_window_bytes = nbytes;
        lap("placement enqueued");
        if ((rc = fpng_amd_band_crc(e, &my_crc->raw_crc, &my_crc->end_offset))) return rc;
        lap("placed, band CRC here");
        // the piece shared with the predecessor is set aside: on the root the predecessor's window is about to land on it
        if (g.head) HIP_TRY(hipMemcpyAsync(dx + o_heads + 16 * (size_t)my_pos, d_window, 16, hipMemcpyDeviceToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(dx + o_crc, my_crc, sizeof(CrcRecord), hipMemcpyHostToDevice, s));
    T_TRY(t->all_gather(t->ctx, dx + o_crc, dx + o_crcs, sizeof(CrcRecord), s));
    rep.collectives++;

    // ---- 4: windows to the root ----
    T_TRY(t->group_begin(t->ctx));
    if (is_root) {
        for (int k = 0; k < world; k++) {
            const int r = order[k];
            if (r == rank || !geo[k].bytes) continue;
            if (geo[k].head) T_TRY(t->recv(t->ctx, dx + o_heads + 16 * (size_t)k, 16, r, s));
            if (geo[k].bytes > geo[k].head) T_TRY(t->recv(t->ctx, d_png + geo[k].off + geo[k].head, geo[k].bytes - geo[k].head, r, s));
            rep.received_in_place += geo[k].bytes - geo[k].head; // (straight to its file offset: nothing is copied again)
            rep.shared_pieces += geo[k].head ? 1u : 0u;
        }
    } else if (have_rows) {
        const Geo &g = geo[my_pos];
        if (g.head) T_TRY(t->send(t->ctx, dx + o_heads + 16 * (size_t)my_pos, 16, root, s));
        if (g.bytes > g.head) T_TRY(t->send(t->ctx, d_window + g.head, g.bytes - g.head, root, s));
        rep.sent_bytes += g.bytes;
    }
    T_TRY(t->group_end(t->ctx));
    lap("windows enqueued");
    if (!is_root) {
        HIP_TRY(hipStreamSynchronize(s));
        lap("done");
        return FPNG_AMD_OK;
    }
    // shared pieces: each band wrote zeros where the other's bits are
    for (int k = 0; k < world; k++)
        if (geo[k].head) launch_or_piece(s, d_png + geo[k].off, dx + o_heads + 16 * (size_t)k);
    HIP_TRY(hipGetLastError());

    // ---- 5: the container ----
    HIP_TRY(hipMemcpyAsync(hx + o_crcs, dx + o_crcs, sizeof(CrcRecord) * world, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    lap("windows and CRC records in");
    const CrcRecord *crcs = (const CrcRecord *)(hx + o_crcs);
    std::vector<uint32_t> raw(world);
    std::vector<uint64_t> ends(world);
    for (int r = 0; r < world; r++) raw[r] = crcs[r].raw_crc, ends[r] = crcs[r].end_offset;
    uint8_t *ht = hx + o_heads; // (the pinned mirror of the heads area is free: head 58 B + tail 20 B)
    if ((rc = fpng_amd_png_head(w, h, c, plan.zlib_size, ht))) return rc;
    fpng_amd_png_tail(plan.adler, fpng_amd_idat_crc_from_bands(raw.data(), ends.data(), (uint32_t)world, plan.zlib_size, plan.adler), ht + 64);
    HIP_TRY(hipMemcpyAsync(d_png, ht, kPngHeaderBytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_png + kPngHeaderBytes + plan.zlib_size - 4, ht + 64, 20, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    *png_size = kPngHeaderBytes + (size_t)plan.zlib_size + kPngTrailerBytes;
    lap("file complete");
    return FPNG_AMD_OK;
}

// ------------------------------------------------------------------------------------------------
// the RCCL transport (librccl.so.1, loaded on first use)
// ------------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    typedef struct { char internal[128]; } UniqueId;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool ok = false;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
constexpr int kNcclUint8 = 1, kNcclUint32 = 3, kNcclSum = 0; // rccl.h: ncclDataType_t / ncclRedOp_t

bool rccl()
{
    std::call_once(g_rccl_once, [] {
        void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        *(void **)&g_rccl.GetUniqueId = sym("ncclGetUniqueId");
        *(void **)&g_rccl.CommInitRank = sym("ncclCommInitRank");
        *(void **)&g_rccl.CommDestroy = sym("ncclCommDestroy");
        *(void **)&g_rccl.AllGather = sym("ncclAllGather");
        *(void **)&g_rccl.AllReduce = sym("ncclAllReduce");
        *(void **)&g_rccl.Send = sym("ncclSend");
        *(void **)&g_rccl.Recv = sym("ncclRecv");
        *(void **)&g_rccl.GroupStart = sym("ncclGroupStart");
        *(void **)&g_rccl.GroupEnd = sym("ncclGroupEnd");
        g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.AllReduce && g_rccl.Send &&
                    g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
    });
    return g_rccl.ok;
}

struct RcclTransport {
    fpng_amd_transport t;
    void *comm = nullptr;
};
int r_all_gather(void *ctx, const void *snd, void *rcv, size_t bytes, void *stream)
{
    return g_rccl.AllGather(snd, rcv, bytes, kNcclUint8, ((RcclTransport *)ctx)->comm, (hipStream_t)stream);
}
int r_all_reduce(void *ctx, void *buf, size_t count, void *stream)
{
    return g_rccl.AllReduce(buf, buf, count, kNcclUint32, kNcclSum, ((RcclTransport *)ctx)->comm, (hipStream_t)stream);
}
int r_group_begin(void *) { return g_rccl.GroupStart(); }
int r_group_end(void *) { return g_rccl.GroupEnd(); }
int r_send(void *ctx, const void *buf, size_t bytes, int peer, void *stream)
{
    return g_rccl.Send(buf, bytes, kNcclUint8, peer, ((RcclTransport *)ctx)->comm, (hipStream_t)stream);
}
int r_recv(void *ctx, void *buf, size_t bytes, int peer, void *stream)
{
    return g_rccl.Recv(buf, bytes, kNcclUint8, peer, ((RcclTransport *)ctx)->comm, (hipStream_t)stream);
}

} // namespace

extern "C" {

int fpng_amd_sharded_last_report(fpng_amd_encoder *e, fpng_amd_sharded_report *report)
{
    if (!e || !report) return fail(FPNG_AMD_ERR_INVALID_ARG, "null argument");
    *report = e->sharded_report;
    return FPNG_AMD_OK;
}

int fpng_amd_rccl_unique_id(uint8_t id[128])
{
    if (!id) return fail(FPNG_AMD_ERR_INVALID_ARG, "null id");
    if (!rccl()) return fail(FPNG_AMD_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    RcclApi::UniqueId u;
    if (g_rccl.GetUniqueId(&u)) return fail(FPNG_AMD_ERR_HIP, "ncclGetUniqueId failed");
    std::memcpy(id, u.internal, 128);
    return FPNG_AMD_OK;
}

int fpng_amd_rccl_transport_create(fpng_amd_transport **out, const uint8_t id[128], int rank, int world, int device)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(FPNG_AMD_ERR_INVALID_ARG, "bad argument");
    *out = nullptr;
    if (!rccl()) return fail(FPNG_AMD_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    RcclTransport *rt = new RcclTransport();
    RcclApi::UniqueId u;
    std::memcpy(u.internal, id, 128);
    if (g_rccl.CommInitRank(&rt->comm, world, u, rank)) {
        delete rt;
        return fail(FPNG_AMD_ERR_HIP, "ncclCommInitRank failed");
    }
    rt->t.ctx = rt, rt->t.rank = rank, rt->t.world = world;
    rt->t.all_gather = r_all_gather, rt->t.all_reduce_sum_u32 = r_all_reduce;
    rt->t.group_begin = r_group_begin, rt->t.group_end = r_group_end, rt->t.send = r_send, rt->t.recv = r_recv;
    *out = &rt->t;
    return FPNG_AMD_OK;
}

void fpng_amd_rccl_transport_destroy(fpng_amd_transport *t)
{
    if (!t) return;
    RcclTransport *rt = (RcclTransport *)t->ctx;
    if (rt->comm) (void)g_rccl.CommDestroy(rt->comm);
    delete rt;
}

} // extern "C"
