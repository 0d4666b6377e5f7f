// Does hipExtAnyOrderLaunch let the second of two independent kernels of ONE stream start before the first has finished on gfx950?
// Two spin kernels of 128 workgroups each (half the chip): in order they take 2 x T, overlapped ~T.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/anyorder tools/probe/anyorder.hip && tools/probe/anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>

__global__ void spin(long long cycles, int *sink)
{
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 100000) *sink = 1;
}

int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const long long cyc = 100000;                     // ~50 us at 2 GHz
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a, s);
            for (int i = 0; i < 8; ++i) {
                hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, cyc, (int *)nullptr);
                if (mode == 0) hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, cyc, (int *)nullptr);
                else hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0, cyc, (int *)nullptr);
            }
            hipEventRecord(b, s);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%s: 8 pairs of 128-workgroup spin kernels: %.1f us per pair\n", mode == 0 ? "in order (hipLaunchKernelGGL)" : mode == 1 ? "second of each pair hipExtAnyOrderLaunch" : "hipExtLaunchKernelGGL, flags 0", best * 1000.f / 8);
    }
    return 0;
}
