// pix_fetch.hip -- how long does ONE pixel-tile fill of the row-sharing convolution kernels take when every CU asks for its own
// at the same time?  A fill = NP pieces (8 pixels x 128 B at a pixel stride of `stride` bytes, LDS-DMA, issued by NW waves of one
// 512-thread workgroup per CU), each workgroup reading its own consecutive pixels of a [npix][stride] tensor.
//   hipcc --offload-arch=gfx950 -O3 -o pix_fetch pix_fetch.hip && ./pix_fetch
// Reports, per (waves issuing, stride, pieces per fill, cache state): cycles from the first issue to vmcnt(0) of the slowest
// wave of the workgroup (median / max over workgroups) and the implied chip-wide rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__device__ __forceinline__ void piece(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}

__global__ void __launch_bounds__(512) k_fill(const char *src, unsigned srcbytes, int stride, int np, int nw, int fills, long long *out)
{
    __shared__ __attribute__((aligned(1024))) char lds[128 * 1024];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, srcbytes, 0x00020000);
    const unsigned ldsbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)lds;
    const int l8 = lane >> 3, lc = lane & 7;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    long long worst = 0;
    for (int f = 0; f < fills; ++f) {
        // fill f of workgroup b: pixels [(b * fills + f) * np * 8, + np * 8)
        const long long pix0 = ((long long)blockIdx.x * fills + f) * np * 8;
        if (wid < nw)
            for (int p = wid; p < np; p += nw) {
                const unsigned voff = (unsigned)((pix0 + p * 8 + l8) * stride + lc * 16);
                piece(rsrc, voff, ldsbase + (p & 63) * 1024);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    worst = t1 - t0;
    if (threadIdx.x == 0) out[blockIdx.x] = worst;
}

int main()
{
    const size_t bytes = 512ull << 20;
    char *src, *flush; long long *out;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    hipMalloc(&flush, 1ull << 30);
    hipMalloc(&out, 256 * sizeof(long long));
    for (int stride : {128, 256, 512})
        for (int np : {19, 37, 74})
            for (int nw : {1, 2, 4, 8})
                for (int fills : {1, 6})
                    for (int warm = 0; warm < 2; ++warm) {
                        if (!warm) hipMemset(flush, warm + np, 1ull << 30);       // evict L2 / Infinity Cache
                        else hipLaunchKernelGGL(k_fill, dim3(256), dim3(512), 0, 0, src, (unsigned)(bytes - 1), stride, np, nw, fills, out);
                        hipDeviceSynchronize();
                        hipLaunchKernelGGL(k_fill, dim3(256), dim3(512), 0, 0, src, (unsigned)(bytes - 1), stride, np, nw, fills, out);
                        hipDeviceSynchronize();
                        std::vector<long long> h(256);
                        hipMemcpy(h.data(), out, 256 * sizeof(long long), hipMemcpyDeviceToHost);
                        std::sort(h.begin(), h.end());
                        const double cyc = (double)h[128];
                        const double us = cyc / 100.0;                              // s_memtime ticks at 100 MHz on gfx950
                        printf("stride %3d  pieces/fill %2d  waves %d  fills %d  %s : median %7.0f max %7.0f ticks (x10 ns) -> %.2f us per fill, %.2f TB/s chip-wide\n", stride, np, nw,
                               fills, warm ? "warm" : "cold", cyc, (double)h[255], us / fills, 256.0 * np * 1024.0 * fills / (us * 1e-6) / 1e12);
                    }
    return 0;
}
