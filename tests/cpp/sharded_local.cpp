// Test-only: fpng_amd_encode_image_sharded() with an IN-PROCESS transport -- `world` ranks as threads of one process, each
// with its own encoder on device 0, exchanging through device-to-device copies.  A 1-GPU box cannot run RCCL with two ranks
// (one rank per device), so this is how the multi-rank logic (plan, windows, shared pieces, CRC combine, stored outcome, root
// != 0, ranks without rows) is exercised on hardware; it also shows what a caller-provided fpng_amd_transport looks like.
#include "fpng_amd.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Post {
    const void *ptr;
    size_t bytes;
    hipEvent_t ready;
    bool taken = false;
};
struct Hub {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool failed = false; // a rank gave up: nobody waits for it any more
    uint8_t *d_slots = nullptr; // world x 8 KiB
    std::map<std::pair<int, int>, std::deque<Post *>> box; // (src, dst) -> posts in order
    bool barrier() // false: a rank failed (or two minutes passed) -- the caller reports a transport error
    {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = gen;
        if (++arrived == world) {
            arrived = 0;
            gen++;
            cv.notify_all();
            return !failed;
        }
        return cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g || failed; }) && !failed;
    }
    void give_up()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            failed = true;
        }
        cv.notify_all();
    }
};
struct Rank {
    Hub *hub;
    int rank;
    std::vector<Post *> sends;                                        // of the open group
    struct Pending { void *buf; size_t bytes; int peer; };
    std::vector<Pending> recvs;
};
constexpr size_t kSlot = 8192;

int l_all_gather(void *ctx, const void *snd, void *rcv, size_t bytes, void *stream)
{
    Rank *r = (Rank *)ctx;
    Hub *h = r->hub;
    if (bytes > kSlot) return 1;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(h->d_slots + r->rank * kSlot, snd, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
    if (!h->barrier()) return 1;
    for (int k = 0; k < h->world; k++)
        if (hipMemcpyAsync((uint8_t *)rcv + k * bytes, h->d_slots + k * kSlot, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    return h->barrier() ? 0 : 1;
}
int l_all_reduce(void *ctx, void *buf, size_t count, void *stream)
{
    Rank *r = (Rank *)ctx;
    Hub *h = r->hub;
    if (count * 4 > kSlot) return 1;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(h->d_slots + r->rank * kSlot, buf, count * 4, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
    if (!h->barrier()) return 1;
    std::vector<uint32_t> sum(count, 0), part(count);
    for (int k = 0; k < h->world; k++) {
        if (hipMemcpy(part.data(), h->d_slots + k * kSlot, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (size_t i = 0; i < count; i++) sum[i] += part[i];
    }
    if (hipMemcpy(buf, sum.data(), count * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return h->barrier() ? 0 : 1;
}
int l_group_begin(void *) { return 0; }
int l_send(void *ctx, const void *buf, size_t bytes, int peer, void *stream)
{
    Rank *r = (Rank *)ctx;
    Post *p = new Post{buf, bytes, nullptr};
    if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(p->ready, (hipStream_t)stream) != hipSuccess) return 1;
    {
        std::lock_guard<std::mutex> lk(r->hub->mu);
        r->hub->box[{r->rank, peer}].push_back(p);
    }
    r->hub->cv.notify_all();
    r->sends.push_back(p);
    return 0;
}
int l_recv(void *ctx, void *buf, size_t bytes, int peer, void *)
{
    ((Rank *)ctx)->recvs.push_back({buf, bytes, peer});
    return 0;
}
int l_group_end_s(Rank *r, hipStream_t s)
{
    Hub *h = r->hub;
    for (auto &q : r->recvs) {
        Post *p = nullptr;
        {
            std::unique_lock<std::mutex> lk(h->mu);
            auto &dq = h->box[{q.peer, r->rank}];
            if (!h->cv.wait_for(lk, std::chrono::seconds(120), [&] { return !dq.empty() || h->failed; }) || dq.empty()) return 1;
            p = dq.front();
            dq.pop_front();
        }
        if (p->bytes != q.bytes) return 1;
        if (hipStreamWaitEvent(s, p->ready, 0) != hipSuccess || hipMemcpyAsync(q.buf, p->ptr, q.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return 1;
        {
            std::lock_guard<std::mutex> lk(h->mu);
            p->taken = true;
        }
        h->cv.notify_all();
    }
    r->recvs.clear();
    for (Post *p : r->sends) { // a sender's buffer is its own again once the receiver has copied it
        std::unique_lock<std::mutex> lk(h->mu);
        if (!h->cv.wait_for(lk, std::chrono::seconds(120), [&] { return p->taken || h->failed; }) || !p->taken) return 1;
        lk.unlock();
        (void)hipEventDestroy(p->ready);
        delete p;
    }
    r->sends.clear();
    return 0;
}
thread_local hipStream_t t_stream = nullptr; // group_end has no stream argument: the rank's encoder stream
int l_group_end(void *ctx) { return l_group_end_s((Rank *)ctx, t_stream); }

} // namespace

// Encodes img (host, w x h x c) as `world` row bands cut at cuts[0..world] (cuts[0] = 0, cuts[world] = h; equal neighbours =
// a rank without rows) by `world` threads; the file comes back from rank `root`.  Returns 0 or the first failing rank's code.
// reports (optional): `world` records, every rank's fpng_amd_sharded_last_report() of this call
extern "C" int shim_sharded_local2(int world, const uint32_t *cuts, const uint8_t *img, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, int root,
                                   uint8_t *out, size_t cap, size_t *size, char *err, size_t err_cap, fpng_amd_sharded_report *reports);
extern "C" int shim_sharded_local(int world, const uint32_t *cuts, const uint8_t *img, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, int root,
                                  uint8_t *out, size_t cap, size_t *size, char *err, size_t err_cap)
{
    return shim_sharded_local2(world, cuts, img, w, h, c, flags, root, out, cap, size, err, err_cap, nullptr);
}
extern "C" int shim_sharded_local2(int world, const uint32_t *cuts, const uint8_t *img, uint32_t w, uint32_t h, uint32_t c, uint32_t flags, int root,
                                   uint8_t *out, size_t cap, size_t *size, char *err, size_t err_cap, fpng_amd_sharded_report *reports)
{
    Hub hub;
    hub.world = world;
    if (hipSetDevice(0) != hipSuccess || hipMalloc(&hub.d_slots, (size_t)world * kSlot) != hipSuccess) return -100;
    std::vector<int> rcs(world, 0);
    std::vector<std::thread> th;
    const size_t bpl = (size_t)w * c;
    std::mutex emu;
    for (int r = 0; r < world; r++)
        th.emplace_back([&, r] {
            (void)hipSetDevice(0);
            fpng_amd_encoder *enc = nullptr;
            int rc = fpng_amd_encoder_create(&enc, 0, nullptr);
            uint8_t *d_rows = nullptr, *d_above = nullptr, *d_png = nullptr;
            const uint32_t y0 = cuts[r], y1 = cuts[r + 1];
            const size_t png_cap = fpng_amd_max_encoded_size(w, h, c) + 64;
            if (!rc && y1 > y0) {
                rc = hipMalloc(&d_rows, (size_t)(y1 - y0) * bpl) != hipSuccess ||
                     hipMemcpy(d_rows, img + (size_t)y0 * bpl, (size_t)(y1 - y0) * bpl, hipMemcpyHostToDevice) != hipSuccess;
                if (!rc && y0) rc = hipMalloc(&d_above, bpl) != hipSuccess || hipMemcpy(d_above, img + (size_t)(y0 - 1) * bpl, bpl, hipMemcpyHostToDevice) != hipSuccess;
            }
            if (!rc && r == root) rc = hipMalloc(&d_png, png_cap) != hipSuccess;
            Rank me{&hub, r};
            fpng_amd_transport t = {&me, r, world, l_all_gather, l_all_reduce, l_group_begin, l_send, l_recv, l_group_end};
            size_t n = 0;
            if (!rc) {
                t_stream = (hipStream_t)fpng_amd_encoder_stream(enc);
                fpng_amd_band b = {d_rows, d_above, w, c, y0, y1, h, 0};
                rc = fpng_amd_encode_image_sharded(enc, &t, &b, flags, root, d_png, png_cap, &n);
                if (rc) {
                    hub.give_up(); // (the other ranks must not wait for this one)
                    std::lock_guard<std::mutex> lk(emu);
                    snprintf(err, err_cap, "rank %d: %s", r, fpng_amd_last_error());
                }
            }
            if (!rc && reports) rc = fpng_amd_sharded_last_report(enc, &reports[r]);
            if (!rc && r == root) {
                *size = n;
                rc = (n > cap) ? -101 : (hipMemcpy(out, d_png, n, hipMemcpyDeviceToHost) != hipSuccess);
            }
            rcs[r] = rc;
            if (rc) hub.give_up(); // (whatever went wrong: the other ranks must not wait for this one)
            (void)hipFree(d_rows), (void)hipFree(d_above), (void)hipFree(d_png);
            if (enc) fpng_amd_encoder_destroy(enc);
        });
    for (auto &t : th) t.join();
    (void)hipFree(hub.d_slots);
    for (int r = 0; r < world; r++)
        if (rcs[r]) return rcs[r];
    return 0;
}
