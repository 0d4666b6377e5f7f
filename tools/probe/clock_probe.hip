// Probe: rate of clock64() (s_memtime) vs wall_clock64() (100 MHz) under an MFMA-heavy and an idle-ish loop,
// and the MFMA issue rate per SIMD.  build: hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k(long long *out, int iters, float *sink)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    long long t1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    sink[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main()
{
    long long *out, h[2]; float *sink;
    hipMalloc(&out, 16); hipMalloc(&sink, 4096 * 256 * 4);
    for (int blocks : {1, 256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 20000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            const double mfma_per_simd = (double)iters * 4 * ((blocks * 4 + 1023) / 1024 > 0 ? 1 : 1);
            printf("blocks %4d: clock64 %lld ticks, wall %lld ticks (100MHz) -> clock64 rate %.1f MHz; kernel %.3f ms; per-wave MFMA interval %.1f clock64-ticks; chip %.0f TFLOP/s\n",
                   blocks, h[0], h[1], h[0] / (h[1] / 100.0), ms, (double)h[0] / (iters * 4), blocks * 4.0 * iters * 4 * 32768.0 / (ms * 1e-3) / 1e12);
            (void)mfma_per_simd;
        }
    }
    return 0;
}
