// conv_common.h -- device helpers shared by the convolution kernels (conv.hip, conv_rs.hip): MFMA wrappers per element
// type, the inline-asm LDS-DMA instruction and its counted wait.  gfx950 only.
#pragma once
#include "dcf_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---- LDS-DMA helpers (used by k_conv_igemm_dma and k_conv_wgrad3g)
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the same without the compiler-level memory fence: for waits that only guard REGISTERS filled by an inline-asm load (the
// uses are tied to it through "+v" operands of an asm placed behind it)
template <int N> __device__ __forceinline__ void wait_vmcnt_nomem() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N)); }

// One LDS-DMA instruction: 64 lanes x 16 B, lane l from buffer offset voff[l], to LDS bytes [lds_dst, lds_dst + 1024).
// A lane whose offset is outside the descriptor's range has ZEROS written for it (probed on MI355X:
// tools/probe/lds_dma_oob.hip) -- padding, junk rows and masked channels cost one v_cndmask.
// Inline asm on purpose: hipcc counts a *builtin* LDS-DMA as a pending LDS write and drains it with vmcnt(0)
// before the next ds_read, which would serialise the ring; an asm one is ours to count (wait_vmcnt above).
// M0 (the DMA destination base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

template <typename V> __device__ __forceinline__ V opaque(V v) { asm volatile("" : "+v"(v)); return v; }   // stop re-derivation of lane constants

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    // one 16-byte fragment pair = one K=16 MFMA
    __device__ static __forceinline__ void run(const uint4 &a, const uint4 &b, f32x16 &acc)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <> struct Mma<f16_t> {
    __device__ static __forceinline__ void run(const uint4 &a, const uint4 &b, f32x16 &acc)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // one 16-byte fragment pair = four K=2 MFMAs (lane half h owns k = 4h+j of each 8-group;
    // the k permutation is the same for both operands, so the sum is unchanged)
    __device__ static __forceinline__ void run(const uint4 &a, const uint4 &b, f32x16 &acc)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// Epilogue of the implicit-GEMM kernels: v = acc + shift + res ; relu ; v *= (mask > 0), 8 consecutive channels per access
// (after acc_rows8).  Phases instead of one load-compute-store chain per vector: every residual vector of the wave's tiles
// is folded into the accumulators first, then the shift and the ReLU, then every mask vector, then the stores -- loads of a
// phase are independent of each other (behind a store that may alias them, each used to pay its own latency).
// m[j] = output pixel of this lane in position tile j, or < 0; channel of (i, p) = c0 + 32 i + 16 p.
template <typename T, int TN, int TM>
__device__ __forceinline__ void conv_epilogue_add(f32x16 (&acc)[TN][TM], const int (&m)[TM], int c0, int Cn, const T *res)
{
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (m[j] >= 0) {
                    float rr[8];
                    ld8(res + (size_t)m[j] * Cn + c0 + i * 32 + 16 * p, rr);
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[i][j][8 * p + k] += rr[k];
                }
}

template <typename T, int TN, int TM>
__device__ __forceinline__ void conv_epilogue_phases(f32x16 (&acc)[TN][TM], const int (&m)[TM], int c0, int Cn, const float *shift,
                                                     const T *res, const T *mask, int relu, T *y, const float *rowscale = nullptr)
{
    if (res) conv_epilogue_add<T, TN, TM>(acc, m, c0, Cn, res);
    if (shift) {
        // rowscale (fp32 per output pixel, optional): the shift enters as rowscale[m] * shift[c] -- the bias of a Linear layer
        // under a sum over a pixel's rowscale[m] neighbours (fusion fc2)
        float rsv[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) rsv[j] = rowscale ? (m[j] >= 0 ? rowscale[m[j]] : 0.f) : 1.f;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float sh[8];
                ld8(shift + c0 + i * 32 + 16 * p, sh);
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[i][j][8 * p + k] += rowscale ? rsv[j] * sh[k] : sh[k];
            }
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][j][k] = fmaxf(acc[i][j][k], 0.f);
    }
    if (mask) {
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    if (m[j] >= 0) {
                        float mm[8];
                        ld8(mask + (size_t)m[j] * Cn + c0 + i * 32 + 16 * p, mm);
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[i][j][8 * p + k] = mm[k] > 0.f ? acc[i][j][8 * p + k] : 0.f;
                    }
    }
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                if (m[j] >= 0) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = acc[i][j][8 * p + k];
                    st8(y + (size_t)m[j] * Cn + c0 + i * 32 + 16 * p, v);
                }
}

}  // namespace
