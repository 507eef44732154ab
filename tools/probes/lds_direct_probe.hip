// What exactly does global_load_lds do on gfx950?  (tools/probes: not product code.)  hipcc --offload-arch=gfx950 -O3 lds_direct_probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) uint32_t lds_dw;
typedef __attribute__((address_space(1))) uint32_t glb_dw;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const uint32_t *src, uint32_t *dst, int mode)
{
    __shared__ __attribute__((aligned(16))) uint32_t buf[8192];
    const uint32_t t = threadIdx.x, lane = t & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    for (uint32_t i = t; i < 8192; i += blockDim.x) buf[i] = 0xDEAD0000u + i;
    __syncthreads();
    if (mode == 0) { // dword, whole wave: lane l -> buf[wv*64 + l]?
        __builtin_amdgcn_global_load_lds((const glb_dw *)(src + wv * 64 + lane), (lds_dw *)buf + wv * 64, 4, 0, 0);
    } else if (mode == 1) { // dword, two halves, the upper one a slot further on
        if (lane < 32) __builtin_amdgcn_global_load_lds((const glb_dw *)(src + wv * 64 + lane), (lds_dw *)buf + wv * 66, 4, 0, 0);
        else __builtin_amdgcn_global_load_lds((const glb_dw *)(src + wv * 64 + lane), (lds_dw *)buf + wv * 66 + 1, 4, 0, 0);
    } else { // 16 bytes per lane
        __builtin_amdgcn_global_load_lds((const glb_dw *)(src + 4 * (wv * 64 + lane)), (lds_dw *)buf + 4 * wv * 64, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (uint32_t i = t; i < 8192; i += blockDim.x) dst[i] = buf[i];
}
int main()
{
    std::vector<uint32_t> h(8192), o(8192);
    for (int i = 0; i < 8192; i++) h[i] = i;
    uint32_t *d, *e;
    hipMalloc(&d, 8192 * 4), hipMalloc(&e, 8192 * 4);
    hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; mode++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, e, mode);
        hipMemcpy(o.data(), e, 8192 * 4, hipMemcpyDeviceToHost);
        printf("mode %d:", mode);
        for (int i = 0; i < (mode == 2 ? 24 : 12); i++) printf(" %x", o[i]);
        printf(" | at 64:");
        for (int i = 64; i < 72; i++) printf(" %x", o[i]);
        printf(" | at 256:");
        for (int i = 256; i < 264; i++) printf(" %x", o[i]);
        int changed = 0;
        for (int i = 0; i < 8192; i++) changed += o[i] != 0xDEAD0000u + i;
        printf(" | %d words written\n", changed);
    }
    return 0;
}
