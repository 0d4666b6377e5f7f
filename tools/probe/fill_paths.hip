// Probe: how many bytes per clock does ONE CU take into LDS from an L2-resident buffer through (a) LDS-DMA pieces
// (`buffer_load_dwordx4 ... lds`), (b) register staging (`buffer_load_dwordx4` + `ds_write_b128`), (c) both at once -- half the
// waves each?  If (c) is near (a) + (b), a kernel whose loop waits for its LDS fill (conv_rs.hip's small-M kind, conv_wg1.hip)
// could split its operands over the two paths; if (c) is near max(a, b), the limit is shared and there is nothing to split.
// build: hipcc --offload-arch=gfx950 -O2 fill_paths.hip -o fill_paths ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// 512 threads = 8 waves, one workgroup per CU; every wave moves PIECES one-KiB pieces per iteration into its own LDS region.
// mode bit 0: waves 0-3 use DMA, bit 1: waves 4-7 use DMA; otherwise the register path.  A wave with `off` set idles.
template <int PIECES>
__global__ void __launch_bounds__(512) k_fill(const char *src, unsigned window, int iters, int dma_lo, int dma_hi, int idle_lo, int idle_hi, unsigned *sink)
{
    __shared__ __attribute__((aligned(1024))) char lds[8 * PIECES * 2 * 1024];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool dma = wid < 4 ? dma_lo : dma_hi;
    const bool idle = wid < 4 ? idle_lo : idle_hi;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, window, 0x00020000);
    const unsigned ldsbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds + wid * PIECES * 2 * 1024;
    // every workgroup walks the same window (as the weight tiles of a layer: L2 hits after the first touch), offset per wave
    unsigned off = ((blockIdx.x * 8 + wid) * (PIECES * 1024u) + lane * 16u) % window;
    unsigned acc = 0;
    if (!idle) {
        for (int it = 0; it < iters; ++it) {
            const unsigned slot = ldsbase + (it & 1) * PIECES * 1024;
            if (dma) {
#pragma unroll
                for (int p = 0; p < PIECES; ++p) {
                    unsigned keep;
                    const unsigned o = (off + p * 1024u) % window;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(o), "s"(rsrc), "s"(slot + p * 1024) : "memory");
                }
                // one iteration of pieces stays in flight
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            } else {
                u32x4 v[PIECES];
#pragma unroll
                for (int p = 0; p < PIECES; ++p) v[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((off + p * 1024u) % window), 0, 0);
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
                    *reinterpret_cast<u32x4 *>(lds + (slot - ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds)) + p * 1024 + lane * 16) = v[p];
            }
            off = (off + 256u * 8u * PIECES * 1024u) % window;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    acc = *reinterpret_cast<unsigned *>(lds + (threadIdx.x * 16) % (8 * PIECES * 2 * 1024));
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int PIECES>
static void run_modes(const char *src, unsigned window, unsigned *sink, const char *what)
{
    const int iters = 3200 / PIECES;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char *name; int dlo, dhi, ilo, ihi; int waves; } modes[] = {
        {"LDS-DMA, 8 waves", 1, 1, 0, 0, 8},
        {"LDS-DMA, 4 waves (others idle)", 1, 1, 0, 1, 4},
        {"register staging, 8 waves", 0, 0, 0, 0, 8},
        {"4 waves LDS-DMA + 4 waves register staging", 1, 0, 0, 0, 8},
    };
    for (auto &m : modes) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k_fill<PIECES>), dim3(256), dim3(512), 0, 0, src, window, iters, m.dlo, m.dhi, m.ilo, m.ihi, sink);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)m.waves * PIECES * 1024.0 * iters;     // per CU
        // a wave keeps up to 2 iterations of pieces in flight (DMA: counted wait; registers: one iteration)
        printf("%-10s %2d pieces/wave/iteration (<= %3d KiB in flight per CU)  %-44s %7.1f GB/s per CU  %5.1f B/clk  (chip %.2f TB/s)\n", what, PIECES,
               m.waves * PIECES * 2, m.name, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 2.1e9, bytes * 256 / (ms * 1e-3) / 1e12);
    }
}

int main()
{
    char *src; unsigned *sink;
    const size_t big = 1024u << 20;
    (void)hipMalloc(&src, big); (void)hipMalloc(&sink, 64);
    (void)hipMemset(src, 1, big);
    // 2 MiB: resident in every XCD's 4-MiB L2 (what a layer's weights are); 1 GiB: every piece from HBM (what streamed pixels are)
    run_modes<2>(src, 2u << 20, sink, "L2 window");
    run_modes<4>(src, 2u << 20, sink, "L2 window");
    run_modes<8>(src, 2u << 20, sink, "L2 window");
    run_modes<2>(src, (unsigned)big, sink, "HBM stream");
    run_modes<4>(src, (unsigned)big, sink, "HBM stream");
    run_modes<8>(src, (unsigned)big, sink, "HBM stream");
    return 0;
}
