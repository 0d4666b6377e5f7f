// Probe: what a grid-wide barrier costs inside one launch on MI355X (8 XCDs, non-coherent L2s), as a layer chain would use it:
// every workgroup writes a slice of a buffer, releases it at agent scope, meets the others at an atomic sense barrier, acquires and
// reads what ANOTHER workgroup wrote (checked), N times.  Compared with N launches of the same slice work.
// build: hipcc --offload-arch=gfx950 -O2 grid_barrier.hip -o grid_barrier ; run: ./grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void grid_sync(unsigned *count, unsigned *sense, unsigned nwg, unsigned &local_sense)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        local_sense ^= 1u;
        __atomic_thread_fence(__ATOMIC_RELEASE);                       // agent scope: this workgroup's stores leave its L2
        if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sense, local_sense, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(sense, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != local_sense) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

// slice = `kb` KiB per workgroup per round
__global__ void __launch_bounds__(256) k_chain(float *buf, int kb, int rounds, unsigned *count, unsigned *sense, int *bad)
{
    const unsigned nwg = gridDim.x;
    unsigned ls = 0;
    const int per = kb * 256;                       // floats per workgroup
    for (int r = 0; r < rounds; ++r) {
        float *mine = buf + (size_t)(r & 1) * nwg * per + (size_t)blockIdx.x * per;
        for (int i = threadIdx.x; i < per; i += 256) mine[i] = (float)(r * 1000 + blockIdx.x);
        grid_sync(count, sense, nwg, ls);
        const unsigned other = (blockIdx.x + 37u * (r + 1)) % nwg;      // mostly another XCD
        const float *theirs = buf + (size_t)(r & 1) * nwg * per + (size_t)other * per;
        float s = 0.f;
        for (int i = threadIdx.x; i < per; i += 256) s += theirs[i] - (float)(r * 1000 + other);
        if (s != 0.f) atomicAdd(bad, 1);
    }
}

__global__ void __launch_bounds__(256) k_one(float *buf, int kb, int r, int *bad)
{
    const unsigned nwg = gridDim.x;
    const int per = kb * 256;
    float *mine = buf + (size_t)(r & 1) * nwg * per + (size_t)blockIdx.x * per;
    for (int i = threadIdx.x; i < per; i += 256) mine[i] = (float)(r * 1000 + blockIdx.x);
    if (r > 0) {
        const unsigned other = (blockIdx.x + 37u * r) % nwg;
        const float *theirs = buf + (size_t)((r - 1) & 1) * nwg * per + (size_t)other * per;
        float s = 0.f;
        for (int i = threadIdx.x; i < per; i += 256) s += theirs[i] - (float)((r - 1) * 1000 + other);
        if (s != 0.f) atomicAdd(bad, 1);
    }
}

int main()
{
    const int nwg = 256, rounds = 200;
    for (int kb : {1, 8, 64}) {
        float *buf; unsigned *sync; int *bad, hbad = 0;
        hipMalloc(&buf, (size_t)2 * nwg * kb * 1024); hipMalloc(&sync, 8); hipMalloc(&bad, 4);
        hipMemset(sync, 0, 8); hipMemset(bad, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms_chain = 0, ms_launch = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chain, dim3(nwg), dim3(256), 0, 0, buf, kb, rounds, sync, sync + 1, bad);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_chain, e0, e1);
        }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_one, dim3(nwg), dim3(256), 0, 0, buf, kb, r, bad);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_launch, e0, e1);
        }
        hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
        printf("%3d KiB per workgroup and round (%.1f MB per round): one launch with grid barriers %.2f us per round | one launch per round %.2f us | stale reads %d\n",
               kb, nwg * kb / 1024.0, ms_chain * 1e3 / rounds, ms_launch * 1e3 / rounds, hbad);
        hipFree(buf); hipFree(sync); hipFree(bad);
    }
    return 0;
}
