// stream_probe.hip -- measurement helper, NOT part of libfpng_amd.so (it lived there as fpng_amd_calibration_stream until round 6):
// streams a buffer of known size with the access widths the encoder uses, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be
// converted to bytes on gfx950 (MI355X_MICROARCH.md, HBM section: the counters are not in bytes for every width), and so that the
// probes of the memory system (tools/mall_probe.py, tools/mix_probe.py) have a plain reader and a plain writer.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/libstream_probe.so tools/probes/stream_probe.hip
// (tools/probes/stream_probe.py builds it on first use and binds the one entry point.)
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace {
constexpr int kBlock = 256;
template <typename T> __global__ __launch_bounds__(kBlock) void calib_read_kernel(const T *src, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        if constexpr (sizeof(T) == 16)
            acc ^= src[i].x ^ src[i].w;
        else
            acc ^= (uint32_t)src[i];
    }
    if (acc == 0x12345678u) *sink = acc; // keep the loads alive
}
template <typename T> __global__ __launch_bounds__(kBlock) void calib_write_kernel(T *dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        if constexpr (sizeof(T) == 16)
            dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
        else
            dst[i] = (T)i;
    }
}
uint32_t *g_sink = nullptr;
} // namespace

// write = 0: read `bytes` of d_buf once; 1: write them once; lane_bytes 4 or 16; enqueued on `stream` (0 = the null stream), not waited for.
extern "C" int stream_probe(void *stream, int write, unsigned lane_bytes, void *d_buf, size_t bytes)
{
    if (!d_buf || (lane_bytes != 4 && lane_bytes != 16) || ((uintptr_t)d_buf & 15)) return -1;
    if (!g_sink && hipMalloc(&g_sink, 64) != hipSuccess) return -3;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(256 * 32), block(kBlock); // (every wave slot of the chip taken: the write stream needs that to reach its rate)
    if (!write && lane_bytes == 4) hipLaunchKernelGGL(calib_read_kernel<uint32_t>, grid, block, 0, s, (const uint32_t *)d_buf, bytes / 4, g_sink);
    if (!write && lane_bytes == 16) hipLaunchKernelGGL(calib_read_kernel<uint4>, grid, block, 0, s, (const uint4 *)d_buf, bytes / 16, g_sink);
    if (write && lane_bytes == 4) hipLaunchKernelGGL(calib_write_kernel<uint32_t>, grid, block, 0, s, (uint32_t *)d_buf, bytes / 4);
    if (write && lane_bytes == 16) hipLaunchKernelGGL(calib_write_kernel<uint4>, grid, block, 0, s, (uint4 *)d_buf, bytes / 16);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
