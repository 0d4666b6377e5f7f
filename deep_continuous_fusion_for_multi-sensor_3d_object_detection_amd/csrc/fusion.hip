// fusion.hip -- the gather/scatter half of the continuous-fusion layer (SURVEY.md App. D;
// the reference leaves it as a TODO at /root/reference/model.py:199-203).
//
// The layer  x_s += sum_k fc2(relu(fc1([F(u_k,v_k); dx,dy,z])))  is evaluated as
//   (1) fp[n][Cf]  = bilinear sample of the camera map at every valid point   (this file)
//   (2) P[n][Cb]   = fp . W1f^T                                  (1x1 conv GEMM, conv.hip)
//   (3) hsum[p][c] = sum_k relu(P[idx_k][c] + W1d[c].(dx,dy,z) + b1[c])        (this file)
//   (4) x_s       += hsum . W2^T + cnt*b2                        (1x1 conv GEMM + bias row)
// which is the same arithmetic re-associated: the F-dependent half of fc1 is per POINT, not
// per (pixel, neighbour), and fc2 is linear so it commutes with the K-sum.  Steps (1),(3)
// are gathers of whole contiguous channel rows (coalesced 8-16 B per lane); their backward
// passes are scatter-adds with fp32 atomics.
#include <stdlib.h>

#include <algorithm>

#include "dcf_common.h"

namespace {

struct Taps {
    int x0, x1, y0, y1;
    float w00, w01, w10, w11;
};

// ix = u/4 - 0.5, iy = v/4 - 0.5 on the stride-4 map, border clamp (oracle/model_ref.py bilinear_sample)
__device__ __forceinline__ Taps make_taps(float u, float v, int Hf, int Wf)
{
    const float ix = u * 0.25f - 0.5f, iy = v * 0.25f - 0.5f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx = ix - x0f, wy = iy - y0f;
    Taps t;
    const int x0 = (int)x0f, y0 = (int)y0f;
    t.x0 = min(max(x0, 0), Wf - 1); t.x1 = min(max(x0 + 1, 0), Wf - 1);
    t.y0 = min(max(y0, 0), Hf - 1); t.y1 = min(max(y0 + 1, 0), Hf - 1);
    t.w00 = (1.f - wy) * (1.f - wx); t.w01 = (1.f - wy) * wx;
    t.w10 = wy * (1.f - wx); t.w11 = wy * wx;
    return t;
}

// Batched launches (grid.y = frame of the batch): element strides between the frames' tensors; all zero for a single frame.
struct FrameStride {
    int64_t a, b, c, d, e;
};

template <typename T>
__global__ void __launch_bounds__(256) k_point_sample_fwd(const T *fmap, int Hf, int Wf, int C4, const float *uv, const int *count, int n_max, T *fp,
                                                          FrameStride fs)
{
    fmap += blockIdx.y * fs.a; uv += blockIdx.y * fs.b; count += blockIdx.y * fs.c; fp += blockIdx.y * fs.d;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(*count, n_max);
    const int64_t p = e / C4;
    if (p >= n) {                               // rows past the frame's point count are written as zeros (no pre-zeroed output)
        if (p < n_max) st4(fp + e * 4, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    const int c = (int)(e - p * C4) * 4;
    const Taps t = make_taps(uv[2 * p], uv[2 * p + 1], Hf, Wf);
    const int C = C4 * 4;
    const float4 a = ld4(fmap + ((int64_t)t.y0 * Wf + t.x0) * C + c), b = ld4(fmap + ((int64_t)t.y0 * Wf + t.x1) * C + c);
    const float4 cc = ld4(fmap + ((int64_t)t.y1 * Wf + t.x0) * C + c), d = ld4(fmap + ((int64_t)t.y1 * Wf + t.x1) * C + c);
    float4 o;
    o.x = a.x * t.w00 + b.x * t.w01 + cc.x * t.w10 + d.x * t.w11;
    o.y = a.y * t.w00 + b.y * t.w01 + cc.y * t.w10 + d.y * t.w11;
    o.z = a.z * t.w00 + b.z * t.w01 + cc.z * t.w10 + d.z * t.w11;
    o.w = a.w * t.w00 + b.w * t.w01 + cc.w * t.w10 + d.w * t.w11;
    st4(fp + e * 4, o);
}

__device__ __forceinline__ void atomic_add4(float *p, float4 v, float w)
{
    atomicAdd(p + 0, v.x * w); atomicAdd(p + 1, v.y * w); atomicAdd(p + 2, v.z * w); atomicAdd(p + 3, v.w * w);
}

// One channel per lane: a wave-instruction's atomics then cover whole contiguous channel rows
// (256 B per 64 channels), the shape the memory-side atomic units run at full rate on.
template <typename T>
__global__ void __launch_bounds__(256) k_point_sample_bwd(const T *gfp, int Hf, int Wf, int C, const float *uv, const int *count, int n_max,
                                                          float *gfmap, FrameStride fs)
{
    gfp += blockIdx.y * fs.a; uv += blockIdx.y * fs.b; count += blockIdx.y * fs.c; gfmap += blockIdx.y * fs.d;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(*count, n_max);
    const int64_t p = e / C;
    if (p >= n) return;
    const int c = (int)(e - p * C);
    const Taps t = make_taps(uv[2 * p], uv[2 * p + 1], Hf, Wf);
    const float g = DT<T>::ld(gfp + e);
    atomicAdd(gfmap + ((int64_t)t.y0 * Wf + t.x0) * C + c, g * t.w00);
    atomicAdd(gfmap + ((int64_t)t.y0 * Wf + t.x1) * C + c, g * t.w01);
    atomicAdd(gfmap + ((int64_t)t.y1 * Wf + t.x0) * C + c, g * t.w10);
    atomicAdd(gfmap + ((int64_t)t.y1 * Wf + t.x1) * C + c, g * t.w11);
}

struct FuseGeom {
    int h, w, stride, K;
    float xs, xo, ys, yo;
};

__device__ __forceinline__ void pixel_centre(const FuseGeom &g, int i, int j, float &X, float &Y)
{
    const float s = (float)g.stride;
    X = __fdiv_rn(__fsub_rn(__fmul_rn((float)i + 0.5f, s), g.xo), g.xs);
    Y = __fdiv_rn(__fsub_rn(__fmul_rn((float)j + 0.5f, s), g.yo), g.ys);
}

template <typename T>
__global__ void __launch_bounds__(256) k_fusion_gather_fwd(const T *P, const float *xyz, const int *idx, FuseGeom g, const float *w1d,
                                                           const float *b1, int C4, T *hsum, float *cnt, FrameStride fs)
{
    P += blockIdx.y * fs.a; xyz += blockIdx.y * fs.b; idx += blockIdx.y * fs.c; hsum += blockIdx.y * fs.d; cnt += blockIdx.y * fs.e;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = g.h * g.w;
    const int64_t p = e / C4;
    if (p >= hw) return;
    const int c = (int)(e - p * C4) * 4;
    const int C = C4 * 4;
    float X, Y;
    pixel_centre(g, (int)(p / g.w), (int)(p % g.w), X, Y);
    float wd[4][3], bb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wd[q][0] = w1d[(c + q) * 3]; wd[q][1] = w1d[(c + q) * 3 + 1]; wd[q][2] = w1d[(c + q) * 3 + 2];
        bb[q] = b1[c + q];
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int nv = 0;
    for (int k = 0; k < g.K; ++k) {
        const int id = idx[(int64_t)k * hw + p];
        if (id < 0) continue;
        ++nv;
        const float dx = xyz[3 * id] - X, dy = xyz[3 * id + 1] - Y, dz = xyz[3 * id + 2];
        const float4 pv = ld4(P + (int64_t)id * C + c);
        const float pp[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float pre = pp[q] + (wd[q][0] * dx + wd[q][1] * dy + wd[q][2] * dz) + bb[q];
            acc[q] += fmaxf(pre, 0.f);
        }
    }
    st4(hsum + e * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
    if (c == 0) cnt[p] = (float)nv;
}

// backward: a thread owns one channel of a RUN of consecutive BEV pixels.  Neighbouring pixels mostly
// share their nearest points, so the per-slot (point id, partial sum) is kept in registers and flushed
// with one atomic only when the id changes: in sparse regions thousands of same-address atomics
// collapse into a few.  Each flush of a wave is one contiguous channel-row segment.
template <typename T>
__global__ void __launch_bounds__(256) k_fusion_gather_bwd(const T *P, const float *xyz, const int *idx, FuseGeom g, const float *w1d,
                                                           const float *b1, int C, const T *ghsum, float *gP, float *gw1d, float *gb1,
                                                           int chunk)
{
    extern __shared__ float sm[];  // [C][4]: gw1d x3, gb1
    for (int i = threadIdx.x; i < C * 4; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = g.h * g.w;
    const int64_t group = t / C;
    const int c = (int)(t - group * C);
    const int64_t p_lo = group * chunk;
    if (p_lo < hw) {
        const int p_hi = (int)min((int64_t)hw, p_lo + chunk);
        const float w0 = w1d[c * 3], w1 = w1d[c * 3 + 1], w2 = w1d[c * 3 + 2], bb = b1[c];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;
        constexpr int KMAX = 8;
        int cur_id[KMAX];
        float cur_acc[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { cur_id[k] = -1; cur_acc[k] = 0.f; }
        for (int p = (int)p_lo; p < p_hi; ++p) {
            float X, Y;
            pixel_centre(g, p / g.w, p % g.w, X, Y);
            const float gg = DT<T>::ld(ghsum + (int64_t)p * C + c);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < g.K) {
                    const int id = idx[(int64_t)k * hw + p];
                    if (id != cur_id[k]) {
                        if (cur_id[k] >= 0 && cur_acc[k] != 0.f) atomicAdd(gP + (int64_t)cur_id[k] * C + c, cur_acc[k]);
                        cur_id[k] = id;
                        cur_acc[k] = 0.f;
                    }
                    if (id >= 0) {
                        const float dx = xyz[3 * id] - X, dy = xyz[3 * id + 1] - Y, dz = xyz[3 * id + 2];
                        const float pre = DT<T>::ld(P + (int64_t)id * C + c) + (w0 * dx + w1 * dy + w2 * dz) + bb;
                        const float d = pre > 0.f ? gg : 0.f;
                        cur_acc[k] += d;
                        a0 += d * dx; a1 += d * dy; a2 += d * dz; ab += d;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (cur_id[k] >= 0 && cur_acc[k] != 0.f) atomicAdd(gP + (int64_t)cur_id[k] * C + c, cur_acc[k]);
        atomicAdd(&sm[c * 4 + 0], a0);
        atomicAdd(&sm[c * 4 + 1], a1);
        atomicAdd(&sm[c * 4 + 2], a2);
        atomicAdd(&sm[c * 4 + 3], ab);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        atomicAdd(&gw1d[i * 3 + 0], sm[i * 4 + 0]);
        atomicAdd(&gw1d[i * 3 + 1], sm[i * 4 + 1]);
        atomicAdd(&gw1d[i * 3 + 2], sm[i * 4 + 2]);
        atomicAdd(&gb1[i], sm[i * 4 + 3]);
    }
}

// Same backward with the pixel loop software-pipelined for a compile-time K: the loads of pixel p+2's neighbour ids /
// gradient row and of pixel p+1's point rows are in flight while pixel p is evaluated, so an iteration exposes about
// one memory latency instead of the dependent chain idx -> (xyz, P) -> use.  Same arithmetic and flush order as above.
template <typename T, int KT>
__global__ void __launch_bounds__(256) k_fusion_gather_bwd_pipe(const T *__restrict__ P, const float *__restrict__ xyz, const int *__restrict__ idx, FuseGeom g,
                                                                const float *__restrict__ w1d, const float *__restrict__ b1, int C, const T *__restrict__ ghsum,
                                                                float *gP, float *gw1d, float *gb1, int chunk)
{
    extern __shared__ float sm[];  // [C][4]: gw1d x3, gb1
    for (int i = threadIdx.x; i < C * 4; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = g.h * g.w;
    const int64_t group = t / C;
    const int c = (int)(t - group * C);
    const int64_t p_lo = group * chunk;
    if (p_lo < hw) {
        const int p_hi = (int)min((int64_t)hw, p_lo + chunk);
        const float w0 = w1d[c * 3], w1 = w1d[c * 3 + 1], w2 = w1d[c * 3 + 2], bb = b1[c];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;
        int cur_id[KT];
        float cur_acc[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) { cur_id[k] = -1; cur_acc[k] = 0.f; }
        // stage I: ids + gradient of a pixel; stage V: its point rows
        int idI[KT], idV[KT];
        float ggI, ggV = 0.f;
        float pv[KT], px[KT], py[KT], pz[KT];
        auto load_I = [&](int p) {
            const int pc = min(p, p_hi - 1);              // past the end: a harmless re-read, never used
#pragma unroll
            for (int k = 0; k < KT; ++k) idI[k] = idx[(int64_t)k * hw + pc];
            ggI = DT<T>::ld(ghsum + (int64_t)pc * C + c);
        };
        auto load_V = [&]() {                             // consumes stage I
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int id = idI[k];
                idV[k] = id;
                const int ic = max(id, 0);
                pv[k] = DT<T>::ld(P + (int64_t)ic * C + c);
                px[k] = xyz[3 * ic]; py[k] = xyz[3 * ic + 1]; pz[k] = xyz[3 * ic + 2];
            }
            ggV = ggI;
        };
        load_I((int)p_lo);
        load_V();
        load_I((int)p_lo + 1);
        for (int p = (int)p_lo; p < p_hi; ++p) {
            // take pixel p's operands out of stage V, then refill the pipeline before the arithmetic
            int id[KT];
            float v[KT], x[KT], y[KT], z[KT];
            const float gg = ggV;
#pragma unroll
            for (int k = 0; k < KT; ++k) { id[k] = idV[k]; v[k] = pv[k]; x[k] = px[k]; y[k] = py[k]; z[k] = pz[k]; }
            load_V();                                      // pixel p+1
            load_I(p + 2);
            float X, Y;
            pixel_centre(g, p / g.w, p % g.w, X, Y);
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (id[k] != cur_id[k]) {
                    if (cur_id[k] >= 0 && cur_acc[k] != 0.f) atomicAdd(gP + (int64_t)cur_id[k] * C + c, cur_acc[k]);
                    cur_id[k] = id[k];
                    cur_acc[k] = 0.f;
                }
                if (id[k] >= 0) {
                    const float dx = x[k] - X, dy = y[k] - Y, dz = z[k];
                    const float pre = v[k] + (w0 * dx + w1 * dy + w2 * dz) + bb;
                    const float d = pre > 0.f ? gg : 0.f;
                    cur_acc[k] += d;
                    a0 += d * dx; a1 += d * dy; a2 += d * dz; ab += d;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KT; ++k)
            if (cur_id[k] >= 0 && cur_acc[k] != 0.f) atomicAdd(gP + (int64_t)cur_id[k] * C + c, cur_acc[k]);
        atomicAdd(&sm[c * 4 + 0], a0);
        atomicAdd(&sm[c * 4 + 1], a1);
        atomicAdd(&sm[c * 4 + 2], a2);
        atomicAdd(&sm[c * 4 + 3], ab);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        atomicAdd(&gw1d[i * 3 + 0], sm[i * 4 + 0]);
        atomicAdd(&gw1d[i * 3 + 1], sm[i * 4 + 1]);
        atomicAdd(&gw1d[i * 3 + 2], sm[i * 4 + 2]);
        atomicAdd(&gb1[i], sm[i * 4 + 3]);
    }
}

// Backward driven by the INVERSE of the KNN map (dcf_fusion_invert): the (pixel, point) pairs sorted by point.  A wave
// takes a slice of 128 consecutive pairs, lane = channel (+64 j): every pair is one coalesced row read of the gradient
// and one (L1-resident, the pairs of a point are adjacent) row read of P, all independent of each other -- no
// idx -> point -> row chain, so a wave keeps 4 pairs' loads in flight and the kernel runs at memory rate instead of
// one exposed latency per pixel.  A point's sum is flushed once per slice it appears in (fp32 atomics only there).
// 16 waves per workgroup: the number of workgroups is capped (each ends with C x 4 same-address atomics on dW1d / db1), so
// the waves that hide the gather latency have to come from inside the workgroup
constexpr int FGI_THREADS = 1024;
constexpr int FG_NSLOT = 16;          // copies of the dW1d / db1 accumulators in the workspace (see the end of the kernel)
// EXCL: instead of slices of pairs, a wave owns a range of POINTS (all pairs of its points: start[p0] .. start[p1]), so every
// row of gP has exactly one writer: it is stored whole in the compute type (zeros for points without pairs) -- no zero-fill
// of an fp32 accumulator before, no float atomics, no cast kernel after.  e_begin then points at start[0] of the map, SL is
// the number of point rows.
template <typename T, int CJ, bool EXCL>
__global__ void __launch_bounds__(CJ >= 4 ? FGI_THREADS / 2 : FGI_THREADS) k_fusion_gather_bwd_inv(const T *__restrict__ P, const float *__restrict__ xyz, const int *__restrict__ e_begin, const int *__restrict__ e_end,
                                                               const int *__restrict__ ent_pix, const int *__restrict__ ent_pt, FuseGeom g,
                                                               const float *__restrict__ w1d, const float *__restrict__ b1, int C,
                                                               const T *__restrict__ ghsum, void *gPv, float *gw1d, float *gb1, int SL, float *part,
                                                               FrameStride fs, float *bws = nullptr, int nslots = 0)
{
    // batched launch: frame blockIdx.y -- P / gP rows (a), xyz (b), the map's start segment (c: e_begin and e_end), ghsum (d)
    P += blockIdx.y * fs.a; xyz += blockIdx.y * fs.b; e_begin += blockIdx.y * fs.c; e_end += blockIdx.y * fs.c; ghsum += blockIdx.y * fs.d;
    // DIRECT (bws != null, not EXCL): gP is in the compute type and zero on entry.  A point whose pairs all sit in one slice has one
    // writer: its row is stored whole.  A point whose run crosses slice boundaries is summed in an fp32 row of the workspace
    // (row = the slice its run starts in -- that slice's last point, so a row serves one point) by the slices of its run; each
    // takes a ticket once its atomics are acknowledged and the last one reads the row back (atomic exchanges, which also leave
    // it zero for the next launch) and stores it.  No fp32 accumulator the size of gP to zero before and to cast after.
    const bool direct = !EXCL && bws != nullptr;
    float *gP = reinterpret_cast<float *>(gPv) + ((EXCL || direct) ? 0 : blockIdx.y * fs.a);
    T *gPt = reinterpret_cast<T *>(gPv) + ((EXCL || direct) ? blockIdx.y * fs.a : 0);
    float *bslot = bws + (size_t)blockIdx.y * nslots * (C + 1);
    unsigned *bcnt = reinterpret_cast<unsigned *>(bslot + (size_t)nslots * C);
    constexpr int U = 8;
    extern __shared__ float sm[];  // [C][4]: gw1d x3, gb1
    for (int i = threadIdx.x; i < C * 4; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * (blockDim.x >> 6);   // the grid is capped: a wave takes slices wave, wave + nwaves, ...
    const int PP = EXCL ? (SL + nwaves - 1) / nwaves : 0;       // points per wave
    const int xp0 = wave * PP, xp1 = min(xp0 + PP, SL);
    const int E0 = EXCL ? (xp0 < xp1 ? e_begin[xp0] : 0) : *e_begin, E = EXCL ? (xp0 < xp1 ? e_begin[xp1] : 0) : *e_end;
    const int SLICE = EXCL ? max(E - E0, 1) : SL;             // EXCL: one "slice" = all pairs of the wave's points
    int next_row = xp0;
    auto store_row = [&](int pt, const float (&acc)[CJ]) {
#pragma unroll
        for (int j = 0; j < CJ; ++j) DT<T>::st(gPt + (int64_t)pt * C + lane + 64 * j, acc[j]);
    };
    auto shared_row = [&](int pt, const float (&acc)[CJ], int e0) {
        const int rs = e_begin[pt], re = e_begin[pt + 1];             // the point's run of pairs
        const int s_first = (rs - e0) / SL, s_last = (re - 1 - e0) / SL;
        float *row = bslot + (size_t)s_first * C;
#pragma unroll
        for (int j = 0; j < CJ; ++j)
            if (acc[j] != 0.f) atomicAdd(row + lane + 64 * j, acc[j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&bcnt[s_first], 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t == (unsigned)(s_last - s_first)) {                      // the last of the run's s_last - s_first + 1 slices
            float tot[CJ];
#pragma unroll
            for (int j = 0; j < CJ; ++j) tot[j] = atomicExch(row + lane + 64 * j, 0.f);
            store_row(pt, tot);
            if (lane == 0) atomicExch(&bcnt[s_first], 0u);
        }
    };
    auto zero_rows = [&](int a, int b) {
        for (int rr = a; rr < b; ++rr)
#pragma unroll
            for (int j = 0; j < CJ; ++j) DT<T>::st(gPt + (int64_t)rr * C + lane + 64 * j, 0.f);
    };
    if (EXCL ? (xp0 < xp1) : (E0 + wave * SL < E)) {      // (fewer workgroups = fewer same-address atomics on dW1d / db1)
        float w0[CJ], w1[CJ], w2[CJ], bb[CJ], a0[CJ], a1[CJ], a2[CJ], ab[CJ], cur_acc[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = lane + 64 * j;
            w0[j] = w1d[c * 3]; w1[j] = w1d[c * 3 + 1]; w2[j] = w1d[c * 3 + 2]; bb[j] = b1[c];
            a0[j] = a1[j] = a2[j] = ab[j] = cur_acc[j] = 0.f;
        }
      for (int sidx = EXCL ? 0 : wave; EXCL ? (sidx == 0) : (E0 + sidx * SL < E); sidx += nwaves) {
        const int lo = E0 + sidx * SLICE, hi = EXCL ? E : min(E, lo + SLICE);
        int cur_pt = -1;
        bool first_shared = direct && lo > E0 && ent_pt[lo - 1] == ent_pt[lo];   // the slice's first point began in an earlier slice
        auto bcast_i = [](int v, int i) { return __builtin_amdgcn_readlane(v, i); };
        auto bcast_f = [](float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); };
        for (int base = lo; base < hi; base += 64) {
            const int n = min(64, hi - base);
            // lane-parallel preparation of 64 pairs (one per lane): pixel centre (IEEE division, once per pair instead of
            // once per pair and lane), offsets to the point, row offsets; the pair loop below only broadcasts them
            const bool live = lane < n;
            const int l_pix = live ? ent_pix[base + lane] : 0;
            const int l_pt = live ? ent_pt[base + lane] : 0;
            const int pi = l_pix >> 16, pj = l_pix & 0xffff;
            float Xc, Yc;
            pixel_centre(g, pi, pj, Xc, Yc);
            const float l_dx = xyz[3 * l_pt] - Xc, l_dy = xyz[3 * l_pt + 1] - Yc, l_dz = xyz[3 * l_pt + 2];
            const int l_grow = (pi * g.w + pj) * C, l_prow = l_pt * C;
            for (int i0 = 0; i0 < n; i0 += U) {
                int pt[U];
                float dx[U], dy[U], dz[U], gg[U][CJ], pv[U][CJ];
#pragma unroll
                for (int u = 0; u < U; ++u) {               // issue every load of the group first
                    const int i = min(i0 + u, n - 1);        // past the end: a harmless re-read, skipped below
                    pt[u] = bcast_i(l_pt, i);
                    dx[u] = bcast_f(l_dx, i); dy[u] = bcast_f(l_dy, i); dz[u] = bcast_f(l_dz, i);
                    const int grow = bcast_i(l_grow, i), prow = bcast_i(l_prow, i);
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        gg[u][j] = DT<T>::ld(ghsum + grow + lane + 64 * j);
                        pv[u][j] = DT<T>::ld(P + prow + lane + 64 * j);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (i0 + u >= n) break;
                    if (pt[u] != cur_pt) {                   // wave-uniform
                        if (cur_pt >= 0) {
                            if (EXCL) {
                                store_row(cur_pt, cur_acc);
                                next_row = cur_pt + 1;
                            } else if (direct) {
                                if (first_shared) shared_row(cur_pt, cur_acc, E0); else store_row(cur_pt, cur_acc);
                                first_shared = false;
                            } else {
#pragma unroll
                                for (int j = 0; j < CJ; ++j)
                                    if (cur_acc[j] != 0.f) atomicAdd(gP + (int64_t)cur_pt * C + lane + 64 * j, cur_acc[j]);
                            }
                        }
                        if (EXCL) zero_rows(next_row, pt[u]);       // points of this wave that no pixel chose
                        cur_pt = pt[u];
#pragma unroll
                        for (int j = 0; j < CJ; ++j) cur_acc[j] = 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < CJ; ++j) {
                        const float pre = pv[u][j] + (w0[j] * dx[u] + w1[j] * dy[u] + w2[j] * dz[u]) + bb[j];
                        const float d = pre > 0.f ? gg[u][j] : 0.f;
                        cur_acc[j] += d;
                        a0[j] += d * dx[u]; a1[j] += d * dy[u]; a2[j] += d * dz[u]; ab[j] += d;
                    }
                }
            }
        }
        if (cur_pt >= 0) {
            if (EXCL) {
                store_row(cur_pt, cur_acc);
                next_row = cur_pt + 1;
            } else if (direct) {
                const bool last_shared = hi < E && ent_pt[hi] == cur_pt;
                if (first_shared || last_shared) shared_row(cur_pt, cur_acc, E0); else store_row(cur_pt, cur_acc);
            } else {
#pragma unroll
                for (int j = 0; j < CJ; ++j)
                    if (cur_acc[j] != 0.f) atomicAdd(gP + (int64_t)cur_pt * C + lane + 64 * j, cur_acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j) cur_acc[j] = 0.f;
      }
      if (EXCL) zero_rows(next_row, xp1);
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = lane + 64 * j;
            atomicAdd(&sm[c * 4 + 0], a0[j]);
            atomicAdd(&sm[c * 4 + 1], a1[j]);
            atomicAdd(&sm[c * 4 + 2], a2[j]);
            atomicAdd(&sm[c * 4 + 3], ab[j]);
        }
    }
    __syncthreads();
    if (part == nullptr) {                         // no workspace: C x 4 device-scope atomics per workgroup, all on the same addresses
        for (int i = threadIdx.x; i < C; i += blockDim.x) {
            atomicAdd(&gw1d[i * 3 + 0], sm[i * 4 + 0]);
            atomicAdd(&gw1d[i * 3 + 1], sm[i * 4 + 1]);
            atomicAdd(&gw1d[i * 3 + 2], sm[i * 4 + 2]);
            atomicAdd(&gb1[i], sm[i * 4 + 3]);
        }
        return;
    }
    // Slotted reduction: 256 workgroups adding onto the SAME C x 4 addresses serialise per address (~15 of the ~46 us of a big
    // site's launch, measured by leaving the atomics out).  With a workspace every workgroup adds its sums into one of
    // FG_NSLOT copies (16 contenders per address instead of 256), takes a ticket once its atomics have been acknowledged, and
    // the LAST workgroup folds the copies into dW1d / db1 (atomic exchanges: coherent reads that also clear the copies for
    // the next launch).  No fences: everything that crosses workgroups is a device-scope atomic.  (A first version published
    // per-workgroup rows behind an agent-scope release: the release writes back the L2 lines the dP atomics had dirtied --
    // +50 us per launch.)  Workspace = ticket word at float 0, copies [FG_NSLOT][1024] from float 64; zero on entry, left zero.
    const int n4 = C * 4;
    unsigned *ticket = reinterpret_cast<unsigned *>(part);
    float *slots = part + 64;
    float *mine = slots + ((blockIdx.x + blockIdx.y * gridDim.x) % FG_NSLOT) * 1024;
    for (int i = threadIdx.x; i < n4; i += blockDim.x)
        if (sm[i] != 0.f) atomicAdd(&mine[i], sm[i]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x * gridDim.y - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < FG_NSLOT; ++q) tot += atomicExch(&slots[q * 1024 + i], 0.f);
        const int c = i >> 2, q = i & 3;
        if (q < 3) gw1d[c * 3 + q] += tot; else gb1[c] += tot;
    }
    if (threadIdx.x == 0) atomicExch(ticket, 0u);            // ready for the next launch on this stream
}

}  // namespace

// ================================================================== C ABI
static int point_sample_fwd_impl(const char *who, int dtype, const void *fmap, int Hf, int Wf, int Cf, const float *uv, int64_t uv_fstride,
                                 const int32_t *count_dev, int n_max, void *fp, int B, hipStream_t s)
{
    DCF_REQUIRE(fmap && uv && count_dev && fp && Cf % 4 == 0 && B >= 1 && B <= 65535, "%s: bad arguments", who);
    if (n_max == 0) return DCF_OK;
    const int64_t total = (int64_t)n_max * (Cf / 4);
    const FrameStride fs = {(int64_t)Hf * Wf * Cf, uv_fstride, 1, (int64_t)n_max * Cf, 0};
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("point_sample_fwd", (double)B * n_max * (5.0 * Cf * sizeof(T) + 8.0), s, hipLaunchKernelGGL(k_point_sample_fwd<T>, dim3(cdiv(total, 256), B), dim3(256), 0, s, (const T *)fmap, Hf, Wf, Cf / 4, uv, count_dev, n_max, (T *)fp, fs)); })
    return DCF_OK;
}

extern "C" int dcf_point_sample_fwd(int dtype, const void *fmap, int Hf, int Wf, int Cf, const float *uv, const int32_t *count_dev,
                                    int n_max, void *fp, dcf_stream_t stream)
{
    return point_sample_fwd_impl("dcf_point_sample_fwd", dtype, fmap, Hf, Wf, Cf, uv, 0, count_dev, n_max, fp, 1, S(stream));
}

extern "C" int dcf_point_sample_fwd_batch(int dtype, const void *fmap, int Hf, int Wf, int Cf, const float *uv, int64_t uv_fstride,
                                          const int32_t *count_dev, int n_max, void *fp, int B, dcf_stream_t stream)
{
    return point_sample_fwd_impl("dcf_point_sample_fwd_batch", dtype, fmap, Hf, Wf, Cf, uv, uv_fstride, count_dev, n_max, fp, B, S(stream));
}

static int point_sample_bwd_impl(const char *who, int dtype, const void *gfp, int Hf, int Wf, int Cf, const float *uv, int64_t uv_fstride,
                                 const int32_t *count_dev, int n_max, float *gfmap, int B, hipStream_t s)
{
    DCF_REQUIRE(gfp && uv && count_dev && gfmap && Cf % 4 == 0 && B >= 1 && B <= 65535, "%s: bad arguments", who);
    if (n_max == 0) return DCF_OK;
    const int64_t total = (int64_t)n_max * Cf;
    const FrameStride fs = {(int64_t)n_max * Cf, uv_fstride, 1, (int64_t)Hf * Wf * Cf, 0};
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("point_sample_bwd", (double)B * n_max * (Cf * sizeof(T) + 4.0 * Cf * 4.0 + 8.0), s, hipLaunchKernelGGL(k_point_sample_bwd<T>, dim3(cdiv(total, 256), B), dim3(256), 0, s, (const T *)gfp, Hf, Wf, Cf, uv, count_dev, n_max, gfmap, fs)); })
    return DCF_OK;
}

extern "C" int dcf_point_sample_bwd(int dtype, const void *gfp, int Hf, int Wf, int Cf, const float *uv, const int32_t *count_dev,
                                    int n_max, float *gfmap, dcf_stream_t stream)
{
    return point_sample_bwd_impl("dcf_point_sample_bwd", dtype, gfp, Hf, Wf, Cf, uv, 0, count_dev, n_max, gfmap, 1, S(stream));
}

extern "C" int dcf_point_sample_bwd_batch(int dtype, const void *gfp, int Hf, int Wf, int Cf, const float *uv, int64_t uv_fstride,
                                          const int32_t *count_dev, int n_max, float *gfmap, int B, dcf_stream_t stream)
{
    return point_sample_bwd_impl("dcf_point_sample_bwd_batch", dtype, gfp, Hf, Wf, Cf, uv, uv_fstride, count_dev, n_max, gfmap, B, S(stream));
}

static int fusion_gather_fwd_impl(const char *who, int dtype, const void *P, int64_t p_rows, const float *xyz, int64_t xyz_fstride, const int32_t *idx,
                                  int K, int h, int w, int stride, float xs, float xo, float ys, float yo, const float *w1d, const float *b1,
                                  int Cb, void *hsum, float *cnt, int B, hipStream_t s)
{
    DCF_REQUIRE(P && xyz && idx && w1d && b1 && hsum && cnt && Cb % 4 == 0 && K >= 1 && B >= 1 && B <= 65535, "%s: bad arguments", who);
    FuseGeom g;
    g.h = h; g.w = w; g.stride = stride; g.K = K; g.xs = xs; g.xo = xo; g.ys = ys; g.yo = yo;
    const int64_t total = (int64_t)h * w * (Cb / 4);
    const FrameStride fs = {p_rows * Cb, xyz_fstride, (int64_t)K * h * w, (int64_t)h * w * Cb, (int64_t)h * w};
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("fusion_gather_fwd", (double)B * h * w * (K * (4.0 + (double)Cb * sizeof(T)) + Cb * sizeof(T) + 4.0), s, hipLaunchKernelGGL(k_fusion_gather_fwd<T>, dim3(cdiv(total, 256), B), dim3(256), 0, s, (const T *)P, xyz, idx, g, w1d, b1, Cb / 4, (T *)hsum, cnt, fs)); })
    return DCF_OK;
}

extern "C" int dcf_fusion_gather_fwd(int dtype, const void *P, const float *xyz, const int32_t *idx, int K, int h, int w,
                                     int stride, float xs, float xo, float ys, float yo, const float *w1d, const float *b1,
                                     int Cb, void *hsum, float *cnt, dcf_stream_t stream)
{
    return fusion_gather_fwd_impl("dcf_fusion_gather_fwd", dtype, P, 0, xyz, 0, idx, K, h, w, stride, xs, xo, ys, yo, w1d, b1, Cb, hsum, cnt, 1, S(stream));
}

extern "C" int dcf_fusion_gather_fwd_batch(int dtype, const void *P, int64_t p_rows, const float *xyz, int64_t xyz_fstride, const int32_t *idx,
                                           int K, int h, int w, int stride, float xs, float xo, float ys, float yo, const float *w1d,
                                           const float *b1, int Cb, void *hsum, float *cnt, int B, dcf_stream_t stream)
{
    return fusion_gather_fwd_impl("dcf_fusion_gather_fwd_batch", dtype, P, p_rows, xyz, xyz_fstride, idx, K, h, w, stride, xs, xo, ys, yo, w1d, b1, Cb, hsum,
                                  cnt, B, S(stream));
}

extern "C" int dcf_fusion_gather_bwd(int dtype, const void *P, const float *xyz, const int32_t *idx, int K, int h, int w,
                                     int stride, float xs, float xo, float ys, float yo, const float *w1d, const float *b1,
                                     int Cb, const void *ghsum, float *gP, float *gw1d, float *gb1, dcf_stream_t stream)
{
    DCF_REQUIRE(P && xyz && idx && w1d && b1 && ghsum && gP && gw1d && gb1 && Cb % 4 == 0, "dcf_fusion_gather_bwd: bad arguments");
    FuseGeom g;
    g.h = h; g.w = w; g.stride = stride; g.K = K; g.xs = xs; g.xo = xo; g.ys = ys; g.yo = yo;
    hipStream_t s = S(stream);
    DCF_REQUIRE(K <= 8, "dcf_fusion_gather_bwd: K must be <= 8");
    const int64_t hw = (int64_t)h * w;
    // pixels per thread run: swept on cfg2 (div 32..512): ~256k threads is the sweet spot between
    // latency hiding (more, shorter runs) and atomic aggregation (fewer, longer runs)
    static DcfOpt thr_env_o("FUSION_THREADS_K"); const char *thr_env = thr_env_o.str();
    const int64_t threads = (thr_env ? atoi(thr_env) : 256) * 1024ll;
    int chunk = (int)(hw * Cb / threads);
    if (chunk < 8) chunk = 8;
    const int64_t groups = (hw + chunk - 1) / chunk;
    static DcfOpt pipe_env_o("FUSION_PIPE"); const char *pipe_env = pipe_env_o.str();
    const bool pipe = !(pipe_env && atoi(pipe_env) == 0);
#define DCF_FGB(KT_) DCF_LAUNCH_B("fusion_gather_bwd", (double)hw * (K * (4.0 + 2.0 * Cb * sizeof(T)) + Cb * sizeof(T)), s, hipLaunchKernelGGL((k_fusion_gather_bwd_pipe<T, KT_>), dim3(cdiv(groups * Cb, 256)), dim3(256), sizeof(float) * Cb * 4, s, (const T *)P, xyz, idx, g, w1d, b1, Cb, (const T *)ghsum, gP, gw1d, gb1, chunk))
    DCF_DISPATCH_DTYPE(dtype, {
        if (pipe && K == 1) DCF_FGB(1);
        else if (pipe && K == 2) DCF_FGB(2);
        else if (pipe && K == 3) DCF_FGB(3);
        else if (pipe && K == 4) DCF_FGB(4);
        else if (pipe && K == 5) DCF_FGB(5);
        else DCF_LAUNCH_B("fusion_gather_bwd", (double)hw * (K * (4.0 + 2.0 * Cb * sizeof(T)) + Cb * sizeof(T)), s, hipLaunchKernelGGL(k_fusion_gather_bwd<T>, dim3(cdiv(groups * Cb, 256)), dim3(256), sizeof(float) * Cb * 4, s, (const T *)P, xyz, idx, g, w1d, b1, Cb, (const T *)ghsum, gP, gw1d, gb1, chunk));
    })
#undef DCF_FGB
    return DCF_OK;
}

// pairs per wave ("slice"): 128 on the big sites; the coarse sites have few pairs (6.6 k at stride 16) and would otherwise run on
// a few dozen waves, one exposed latency after the other
static int fgi_slice(int max_entries)
{
    int sl = cdiv(cdiv(max_entries, 4096), 16) * 16;
    return sl < 16 ? 16 : (sl > 128 ? 128 : sl);
}

static int fusion_gather_bwd_inv_impl(const char *who, int dtype, const void *P, int64_t p_rows, const float *xyz, int64_t xyz_fstride, const int32_t *e_begin,
                                      const int32_t *e_end, int64_t seg_fstride, const int32_t *ent_pix, const int32_t *ent_pt, int max_entries, int h,
                                      int w, int stride, float xs, float xo, float ys, float yo, const float *w1d, const float *b1, int Cb,
                                      const void *ghsum, void *gP, float *gw1d, float *gb1, void *workspace, int B, hipStream_t s,
                                      void *direct_ws = nullptr)
{
    float *ws = reinterpret_cast<float *>(workspace);
    float *bws = reinterpret_cast<float *>(direct_ws);
    DCF_REQUIRE(P && xyz && e_begin && e_end && ent_pix && ent_pt && w1d && b1 && ghsum && gP && gw1d && gb1, "%s: null pointer", who);
    DCF_REQUIRE(Cb % 64 == 0 && Cb >= 64 && Cb <= 256, "%s: Cb must be 64, 128, 192 or 256 (got %d)", who, Cb);
    DCF_REQUIRE(B >= 1 && B <= 64, "%s: 1..64 frames", who);
    if (max_entries <= 0) return DCF_OK;
    FuseGeom g;
    g.h = h; g.w = w; g.stride = stride; g.K = 0; g.xs = xs; g.xo = xo; g.ys = ys; g.yo = yo;
    const int sl = fgi_slice(max_entries);
    const int waves = cdiv(max_entries, sl);
    static DcfOpt cap_env_o("FUSION_BWD_BLOCKS"); const char *cap_env = cap_env_o.str();
    // swept (one frame per launch): 128 / 256 / 512 / uncapped = 0.52 / 0.34 / 0.36 / 0.44 ms per step; a batched launch shares the cap
    const int cap = std::max((cap_env ? atoi(cap_env) : 256) / B, 32);
    // (256 channels: 4 accumulator sets per lane do not fit the 128 registers of a 1024-thread block -- 512 threads there)
    const int thr = Cb >= 256 ? FGI_THREADS / 2 : FGI_THREADS;
    const int blocks = std::min(cdiv(waves, thr / 64), cap);
    const FrameStride fs = {p_rows * Cb, xyz_fstride, seg_fstride, (int64_t)h * w * Cb, 0};
    // (one profile name per instantiation, as rocprofv3 lists them: the four sites run four different kernels)
#define DCF_FGI(CJ_) DCF_LAUNCH_B("fusion_gather_bwd_inv<" #CJ_ ">", (double)B * max_entries * (8.0 + 2.0 * Cb * sizeof(T)), s, hipLaunchKernelGGL((k_fusion_gather_bwd_inv<T, CJ_, false>), dim3(blocks, B), dim3(thr), sizeof(float) * Cb * 4, s, (const T *)P, xyz, e_begin, e_end, ent_pix, ent_pt, g, w1d, b1, Cb, (const T *)ghsum, gP, gw1d, gb1, sl, ws, fs, bws, waves))
    DCF_DISPATCH_DTYPE(dtype, {
        if (Cb == 64) DCF_FGI(1);
        else if (Cb == 128) DCF_FGI(2);
        else if (Cb == 192) DCF_FGI(3);
        else DCF_FGI(4);
    })
#undef DCF_FGI
    return DCF_OK;
}

extern "C" size_t dcf_fusion_gather_bwd_workspace_bytes(int Cb) { (void)Cb; return ((size_t)FG_NSLOT * 1024 + 64) * sizeof(float); }

extern "C" int dcf_fusion_gather_bwd_inv(int dtype, const void *P, const float *xyz, const int32_t *e_begin, const int32_t *e_end, const int32_t *ent_pix,
                                         const int32_t *ent_pt, int max_entries, int h, int w, int stride, float xs, float xo, float ys,
                                         float yo, const float *w1d, const float *b1, int Cb, const void *ghsum, float *gP, float *gw1d,
                                         float *gb1, void *workspace, dcf_stream_t stream)
{
    return fusion_gather_bwd_inv_impl("dcf_fusion_gather_bwd_inv", dtype, P, 0, xyz, 0, e_begin, e_end, 0, ent_pix, ent_pt, max_entries, h, w, stride, xs, xo,
                                      ys, yo, w1d, b1, Cb, ghsum, gP, gw1d, gb1, workspace, 1, S(stream));
}

extern "C" int dcf_fusion_gather_bwd_inv_batch(int dtype, const void *P, int64_t p_rows, const float *xyz, int64_t xyz_fstride, const int32_t *e_begin,
                                               const int32_t *e_end, int64_t seg_fstride, const int32_t *ent_pix, const int32_t *ent_pt,
                                               int max_entries, int h, int w, int stride, float xs, float xo, float ys, float yo, const float *w1d,
                                               const float *b1, int Cb, const void *ghsum, float *gP, float *gw1d, float *gb1, void *workspace,
                                               int B, dcf_stream_t stream)
{
    return fusion_gather_bwd_inv_impl("dcf_fusion_gather_bwd_inv_batch", dtype, P, p_rows, xyz, xyz_fstride, e_begin, e_end, seg_fstride, ent_pix, ent_pt,
                                      max_entries, h, w, stride, xs, xo, ys, yo, w1d, b1, Cb, ghsum, gP, gw1d, gb1, workspace, B, S(stream));
}

// The same sums with gP in the COMPUTE type (dtype), zero on entry: rows whose pairs sit in one slice are stored whole, rows that
// cross slices are summed in `direct_ws` (dcf_fusion_gather_bwd_direct_workspace_bytes; zero on entry, left zero) and stored by
// the last slice of the run -- no fp32 accumulator the size of gP to fill before the launch and to cast after it.
extern "C" size_t dcf_fusion_gather_bwd_direct_workspace_bytes(int max_entries, int Cb, int B)
{
    if (max_entries <= 0 || B <= 0) return 0;
    return (size_t)B * cdiv(max_entries, fgi_slice(max_entries)) * (Cb + 1) * sizeof(float);
}

extern "C" int dcf_fusion_gather_bwd_direct_batch(int dtype, const void *P, int64_t p_rows, const float *xyz, int64_t xyz_fstride, const int32_t *e_begin,
                                                  const int32_t *e_end, int64_t seg_fstride, const int32_t *ent_pix, const int32_t *ent_pt,
                                                  int max_entries, int h, int w, int stride, float xs, float xo, float ys, float yo, const float *w1d,
                                                  const float *b1, int Cb, const void *ghsum, void *gP, float *gw1d, float *gb1, void *workspace,
                                                  void *direct_ws, int B, dcf_stream_t stream)
{
    DCF_REQUIRE(direct_ws, "dcf_fusion_gather_bwd_direct_batch: null direct_ws");
    return fusion_gather_bwd_inv_impl("dcf_fusion_gather_bwd_direct_batch", dtype, P, p_rows, xyz, xyz_fstride, e_begin, e_end, seg_fstride, ent_pix, ent_pt,
                                      max_entries, h, w, stride, xs, xo, ys, yo, w1d, b1, Cb, ghsum, gP, gw1d, gb1, workspace, B, S(stream), direct_ws);
}

// Same sums with one writer per point row (see k_fusion_gather_bwd_inv<.., EXCL>): gP [n_rows][Cb] in the COMPUTE type, every row
// written (zeros where no pixel chose the point), no zero-fill needed before and no cast after.  start = the map's
// start[g*(n_max+1) ...] (n_rows + 1 entries are read: n_rows <= n_max).
extern "C" int dcf_fusion_gather_bwd_pts(int dtype, const void *P, const float *xyz, const int32_t *start, int n_rows, const int32_t *ent_pix,
                                         const int32_t *ent_pt, int max_entries, int h, int w, int stride, float xs, float xo, float ys, float yo,
                                         const float *w1d, const float *b1, int Cb, const void *ghsum, void *gP, float *gw1d, float *gb1,
                                         dcf_stream_t stream)
{
    DCF_REQUIRE(P && xyz && start && ent_pix && ent_pt && w1d && b1 && ghsum && gP && gw1d && gb1, "dcf_fusion_gather_bwd_pts: null pointer");
    DCF_REQUIRE(Cb % 64 == 0 && Cb >= 64 && Cb <= 256, "dcf_fusion_gather_bwd_pts: Cb must be 64, 128, 192 or 256 (got %d)", Cb);
    if (n_rows <= 0) return DCF_OK;
    FuseGeom g;
    g.h = h; g.w = w; g.stride = stride; g.K = 0; g.xs = xs; g.xo = xo; g.ys = ys; g.yo = yo;
    hipStream_t s = S(stream);
    static DcfOpt cap_env_o("FUSION_BWD_BLOCKS"); const char *cap_env = cap_env_o.str();
    const int cap = cap_env ? atoi(cap_env) : 256;
    const int thr = Cb >= 256 ? FGI_THREADS / 2 : FGI_THREADS;
    const int blocks = std::min(cdiv(n_rows, thr / 64), cap);
#define DCF_FGP(CJ_) DCF_LAUNCH_B("fusion_gather_bwd_pts", (double)max_entries * (8.0 + 2.0 * Cb * sizeof(T)) + (double)n_rows * Cb * sizeof(T), s, hipLaunchKernelGGL((k_fusion_gather_bwd_inv<T, CJ_, true>), dim3(blocks), dim3(thr), sizeof(float) * Cb * 4, s, (const T *)P, xyz, start, start, ent_pix, ent_pt, g, w1d, b1, Cb, (const T *)ghsum, gP, gw1d, gb1, n_rows, (float *)nullptr, FrameStride{0, 0, 0, 0, 0}))
    DCF_DISPATCH_DTYPE(dtype, {
        if (Cb == 64) DCF_FGP(1);
        else if (Cb == 128) DCF_FGP(2);
        else if (Cb == 192) DCF_FGP(3);
        else DCF_FGP(4);
    })
#undef DCF_FGP
    return DCF_OK;
}
