// dma_issue.hip -- how fast can ONE wave issue LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction), by
// how M0 is handled and by how many waves of the CU issue at once?  (Question raised by conv_lc.hip: two loader waves per role
// took ~150-300 cycles per piece even with out-of-range offsets.)
//   hipcc --offload-arch=gfx950 -O3 -o dma_issue dma_issue.hip && ./dma_issue
// Per (mode, waves per workgroup, source): cycles per piece per wave (s_memtime around 64 x 32 pieces, median over workgroups).
// modes: 0 = M0 saved / set / restored around every piece (glds16 of conv_common.h), 1 = M0 set before every piece (clobbered),
//        2 = M0 set once, every piece to the same LDS bytes, 3 = plain buffer_load_dwordx4 into registers,
//        4 = M0 set before every piece, s_nop 0 dropped (NOT safe in general: measures the nop's share only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void piece(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_dst, u32x4 &sink)
{
    if constexpr (MODE == 0) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
    } else if constexpr (MODE == 1) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
    } else if constexpr (MODE == 2) {
        asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc) : "memory");
    } else if constexpr (MODE == 3) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(sink) : "v"(voff), "s"(rsrc));
    } else {
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
    }
}

template <int MODE>
__global__ void __launch_bounds__(512) k_issue(const char *src, unsigned srcbytes, int oob, long long *out, u32x4 *sinkbuf)
{
    __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, srcbytes, 0x00020000);
    const unsigned ldsbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)lds + wid * 8192;
    u32x4 sink = {0, 0, 0, 0};
    if (MODE == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(ldsbase) : "m0");
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            // every workgroup streams the same 256 KiB (L2-resident after the first touch); oob: nothing is fetched
            const unsigned voff = oob ? 0xFFFFFF00u : (unsigned)(((it * 32 + p) & 255) * 1024 + lane * 16);
            piece<MODE>(rsrc, voff, ldsbase + (p & 7) * 1024, sink);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
    if (MODE == 3 && sink.x == 0x12345678u) sinkbuf[threadIdx.x] = sink;
}

template <int MODE> static void run(const char *src, unsigned bytes, long long *out, u32x4 *sinkbuf)
{
    for (int oob = 0; oob < 2; ++oob)
        for (int waves : {1, 2, 4, 8}) {
            hipMemset(out, 0, 256 * 8 * sizeof(long long));
            hipLaunchKernelGGL(k_issue<MODE>, dim3(256), dim3(waves * 64), 0, 0, src, bytes, oob, out, sinkbuf);
            hipLaunchKernelGGL(k_issue<MODE>, dim3(256), dim3(waves * 64), 0, 0, src, bytes, oob, out, sinkbuf);
            hipDeviceSynchronize();
            std::vector<long long> h(256 * 8);
            hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            std::vector<double> v;
            for (int b = 0; b < 256; ++b)
                for (int w = 0; w < waves; ++w) v.push_back((double)h[b * 8 + w] / (64.0 * 32.0));
            std::sort(v.begin(), v.end());
            printf("mode %d  %s  waves/CU %d : %7.1f cycles per piece per wave (median; min %.1f max %.1f) -> %.1f cycles per piece CU-wide\n", MODE,
                   oob ? "out-of-range" : "L2-resident ", waves, v[v.size() / 2], v.front(), v.back(), v[v.size() / 2] / waves);
        }
}

int main()
{
    const unsigned bytes = 256 * 1024;
    char *src; long long *out; u32x4 *sinkbuf;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
    hipMalloc(&out, 256 * 8 * sizeof(long long));
    hipMalloc(&sinkbuf, 512 * sizeof(u32x4));
    run<0>(src, bytes, out, sinkbuf);
    run<1>(src, bytes, out, sinkbuf);
    run<2>(src, bytes, out, sinkbuf);
    run<3>(src, bytes, out, sinkbuf);
    run<4>(src, bytes, out, sinkbuf);
    return 0;
}
