// Standalone replica of the streamed host path's copy pattern with std::thread (no kernels): hipcc tools/pcie_probe3.cpp -o /tmp/p3
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
int main(int argc, char **argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t IN = 132710400, OUT = 58040724;
    const int NB = 8;
    uint8_t *d_in, *d_out;
    hipMalloc(&d_in, IN);
    hipMalloc(&d_out, OUT + 4096);
    uint8_t *h_in = (uint8_t *)malloc(IN), *h_out = (uint8_t *)malloc(OUT + 4096);
    memset(h_in, 1, IN);
    memset(h_out, 1, OUT + 4096);
    hipStream_t su, sd;
    if (variant & 2) { // five more streams created (and used) first, like an encoder's ordering stream + four lanes
        for (int i = 0; i < 5; i++) {
            hipStream_t x;
            hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
            hipMemsetAsync(d_in, 0, 1024, x);
            hipStreamSynchronize(x);
        }
    }
    hipStreamCreateWithFlags(&su, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sd, hipStreamNonBlocking);
    for (int rep = 0; rep < 6; rep++) {
        std::mutex mu;
        std::condition_variable cv;
        int done = 0;
        std::vector<double> tl(NB * 4);
        auto t0 = std::chrono::steady_clock::now();
        auto now = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
        std::thread up([&] {
            if (variant & 1) hipSetDevice(0);
            const size_t n = IN / NB;
            for (int k = 0; k < NB; k++) {
                tl[k * 4] = now();
                hipMemcpyAsync(d_in + k * n, h_in + k * n, n, hipMemcpyHostToDevice, su);
                hipStreamSynchronize(su);
                tl[k * 4 + 1] = now();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    done = k + 1;
                }
                cv.notify_all();
            }
        });
        std::thread down([&] {
            if (variant & 1) hipSetDevice(0);
            const size_t n = (OUT / NB) & ~(size_t)15;
            for (int k = 0; k < NB; k++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return done > k; });
                }
                tl[k * 4 + 2] = now();
                hipMemcpyAsync(h_out + k * n, d_out + k * n, n, hipMemcpyDeviceToHost, sd);
                hipStreamSynchronize(sd);
                tl[k * 4 + 3] = now();
            }
        });
        up.join();
        down.join();
        const double total = now();
        if (rep >= 4) {
            printf("variant %d: total %.0f us:", variant, total);
            for (int k = 0; k < NB; k++) printf(" u%d %.0f-%.0f d%d %.0f-%.0f |", k, tl[k * 4], tl[k * 4 + 1], k, tl[k * 4 + 2], tl[k * 4 + 3]);
            printf("\n");
        }
    }
    return 0;
}
