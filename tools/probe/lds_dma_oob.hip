// Probe: what does an LDS-DMA buffer load write for lanes whose offset is out of range / EXEC-masked?
// build: hipcc --offload-arch=gfx950 -O2 lds_dma_oob.hip -o lds_dma_oob ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const unsigned *src, unsigned nbytes, unsigned *out)
{
    __shared__ __attribute__((aligned(1024))) unsigned lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xABABABABu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    unsigned off = lane * 16;
    if (lane >= 48) off = 0xFFFFFF00u;            // out of range lanes
    unsigned ldsbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned *)lds;
    unsigned keep;
    if (lane < 56)                                  // lanes 56..63 EXEC-masked
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(rsrc), "s"(ldsbase) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main()
{
    unsigned *src, *out, h[512];
    hipMalloc(&src, 4096); hipMalloc(&out, 2048);
    for (int i = 0; i < 512; ++i) h[i] = 0x1000 + i;
    hipMemcpy(src, h, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, 40 * 16, out);   // descriptor covers lanes 0..39 only
    hipMemcpy(h, out, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) if (l < 2 || (l >= 38 && l < 42) || (l >= 46 && l < 50) || l >= 54 && l < 58) printf("lane %2d: %08x %08x %08x %08x\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    printf("tail word 256: %08x\n", h[256]);
    return 0;
}
