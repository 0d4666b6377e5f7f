// elementwise.hip -- the HBM-bound kernels around the convolutions: layout conversion,
// ReLU backward + channel sums, bilinear resize, 3x3/2 max-pool, detection heads
// (softmax pairs + box decode), per-step weight preparation / gradient finalisation
// (folded BatchNorm chain rule) and the fused Adam step.
//
// All activations are NHWC; every kernel moves 8-16 B per lane and is priced against the
// HBM roofline.  Reference lines are cited per kernel.
#include <stdlib.h>

#include "dcf_common.h"

namespace {

// ------------------------------------------------------------------------------------
// NCHW fp32 -> NHWC dtype.  One thread per pixel: coalesced 4-B reads across the wave
// for each channel plane, one contiguous C-vector store per thread.
// (model.py:194 takes the voxel grid as [B,Cz,L,W]; the engine computes in NHWC.)
// ------------------------------------------------------------------------------------
template <typename T, int CV>
__global__ void __launch_bounds__(256) k_nchw_to_nhwc(const float *x, T *y, int C, int64_t HW, int64_t total)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // b*HW + pixel
    if (p >= total) return;
    const int64_t b = p / HW, q = p - b * HW;
    const float *src = x + b * C * HW + q;
    T *dst = y + p * C;
    for (int c0 = 0; c0 < C; c0 += CV) {
        float v[CV];
#pragma unroll
        for (int k = 0; k < CV; ++k) v[k] = src[(int64_t)(c0 + k) * HW];
#pragma unroll
        for (int k = 0; k < CV; k += 4) st4(dst + c0 + k, make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]));
    }
}

// NHWC (compute type) -> NCHW fp32: the way back, for module surfaces that hand activations to torch code in the
// reference's layout (model.py:76-79 returns NCHW stage outputs).  One thread per pixel: channel rows are read 4 at a time,
// the stores of a wave are 64 consecutive pixels of one channel plane.
template <typename T>
__global__ void __launch_bounds__(256) k_nhwc_to_nchw(const T *x, float *y, int C, int64_t HW, int64_t total)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int64_t b = p / HW, q = p - b * HW;
    const T *src = x + p * C;
    float *dst = y + b * C * HW + q;
    for (int c0 = 0; c0 < C; c0 += 4) {
        const float4 v = ld4(src + c0);
        dst[(int64_t)c0 * HW] = v.x; dst[(int64_t)(c0 + 1) * HW] = v.y; dst[(int64_t)(c0 + 2) * HW] = v.z; dst[(int64_t)(c0 + 3) * HW] = v.w;
    }
}

// uint8 NCHW -> x/255 NHWC4 with a 3-pixel zero halo (rows and columns), row pitch (W+8)
// pixels so that rows stay 16-B aligned in bf16.  Image tensor contract: data_import_carla.py:62.
template <typename T>
__global__ void __launch_bounds__(256) k_image_to_nhwc4(const uint8_t *img, T *y, int B, int H, int W)
{
    const int Hp = H + 6, Wp = W + 8;
    const int64_t total = (int64_t)B * Hp * Wp;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const int b = (int)(p / ((int64_t)Hp * Wp));
    const int rem = (int)(p - (int64_t)b * Hp * Wp);
    const int hp = rem / Wp, wp = rem - hp * Wp;
    const int h = hp - 3, w = wp - 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h >= 0 && h < H && w >= 0 && w < W) {
        const int64_t HW = (int64_t)H * W;
        const uint8_t *s = img + (int64_t)b * 3 * HW + (int64_t)h * W + w;
        v.x = (float)s[0] / 255.0f;
        v.y = (float)s[HW] / 255.0f;
        v.z = (float)s[2 * HW] / 255.0f;
    }
    st4(y + p * 4, v);
}

// ------------------------------------------------------------------------------------
// ReLU backward + per-channel sum.  g = gy * (y > 0) written in place; gsum[c] += sum g.
// (ReLU: model.py:21,25,40; the channel sum is dL/d(beta) of the folded BatchNorm.)
// Each thread owns one 4-channel group for its whole grid-stride loop.
// ------------------------------------------------------------------------------------
template <int V, typename T>
__device__ __forceinline__ void ldv(const T *p, float (&v)[V])
{
    if constexpr (V == 8) ld8(p, v);
    else { const float4 t = ld4(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
}

template <int V, typename T>
__device__ __forceinline__ void stv(T *p, const float (&v)[V])
{
    if constexpr (V == 8) st8(p, v);
    else st4(p, make_float4(v[0], v[1], v[2], v[3]));
}

// V channels per thread (8 when the channel count allows); two elements per trip with both elements' loads issued before the
// first store (the tensor is updated in place: behind a store that may alias it the next load would wait for its own trip).
template <typename T, int V>
__global__ void __launch_bounds__(256) k_relu_bwd_chansum(T *gy, const T *y, float *gsum, int64_t nvec, int cgroups, int relu,
                                                          int64_t stride)
{
    extern __shared__ float sm[];  // [cgroups*V]
    for (int i = threadIdx.x; i < cgroups * V; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < stride) {
        const int cg = (int)(t % cgroups);
        float acc[V];
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] = 0.f;
        int64_t e = t;
        for (; e + stride < nvec; e += 2 * stride) {
            float g[2][V], yy[2][V];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                ldv<V>(gy + (e + u * stride) * V, g[u]);
                if (relu) ldv<V>(y + (e + u * stride) * V, yy[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (relu) {
#pragma unroll
                    for (int q = 0; q < V; ++q) g[u][q] = yy[u][q] > 0.f ? g[u][q] : 0.f;
                    stv<V>(gy + (e + u * stride) * V, g[u]);
                }
#pragma unroll
                for (int q = 0; q < V; ++q) acc[q] += g[u][q];
            }
        }
        for (; e < nvec; e += stride) {
            float g[V];
            ldv<V>(gy + e * V, g);
            if (relu) {
                float yy[V];
                ldv<V>(y + e * V, yy);
#pragma unroll
                for (int q = 0; q < V; ++q) g[q] = yy[q] > 0.f ? g[q] : 0.f;
                stv<V>(gy + e * V, g);
            }
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] += g[q];
        }
#pragma unroll
        for (int q = 0; q < V; ++q) atomicAdd(&sm[cg * V + q], acc[q]);
    }
    __syncthreads();
    if (gsum)
        for (int i = threadIdx.x; i < cgroups * V; i += blockDim.x) atomicAdd(&gsum[i], sm[i]);
}

// ------------------------------------------------------------------------------------
// Bilinear resize, NHWC.  align_corners=1: nn.UpsamplingBilinear2d (model.py:149,151);
// align_corners=0: the image FPN of SURVEY.md App. D.  Source index rule = ATen's
// area_pixel_compute_source_index.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(int o, float scale, int align, int in_size, int &i0, int &i1, float &l)
{
    float s;
    if (align) s = scale * (float)o;
    else {
        s = scale * ((float)o + 0.5f) - 0.5f;
        s = s < 0.f ? 0.f : s;
    }
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l = s - (float)i0;
}

__host__ __device__ inline float resize_scale(int in, int out, int align)
{
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

template <typename T>
__global__ void __launch_bounds__(256) k_resize_fwd(const T *x, const T *add, T *y, int B, int Hi, int Wi, int Ho, int Wo, int C4,
                                                    int align, float sh, float sw)
{
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    int64_t p = e / C4;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int b = (int)(p / Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(oh, sh, align, Hi, y0, y1, ly);
    src_index(ow, sw, align, Wi, x0, x1, lx);
    const int C = C4 * 4;
    const T *base = x + (int64_t)b * Hi * Wi * C + c * 4;
    const float4 v00 = ld4(base + ((int64_t)y0 * Wi + x0) * C), v01 = ld4(base + ((int64_t)y0 * Wi + x1) * C);
    const float4 v10 = ld4(base + ((int64_t)y1 * Wi + x0) * C), v11 = ld4(base + ((int64_t)y1 * Wi + x1) * C);
    const float hy = 1.f - ly, hx = 1.f - lx;
    float4 o;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    const int64_t oo = e * 4;
    if (add) {
        const float4 a = ld4(add + oo);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    st4(y + oo, o);
}

// gather-form transpose: input pixel (ih,iw) sums every output pixel that sampled it.
// Candidate output rows form a contiguous window around ih/scale; each is re-derived
// exactly with src_index, so the result is deterministic (no atomics).
// V channels per thread (8 when the channel count allows: 16-byte accesses in the 16-bit types, half the index arithmetic
// per byte -- the search below, not the memory system, is what this kernel waits for)
template <typename T, int V>
__global__ void __launch_bounds__(256) k_resize_bwd(const T *gy, T *gx, int B, int Hi, int Wi, int Ho, int Wo, int CV, int align,
                                                    float sh, float sw)
{
    const int64_t total = (int64_t)B * Hi * Wi * CV;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % CV);
    int64_t p = e / CV;
    const int iw = (int)(p % Wi); p /= Wi;
    const int ih = (int)(p % Hi);
    const int b = (int)(p / Hi);
    const int C = CV * V;
    const float inv_h = sh > 0.f ? 1.f / sh : (float)Ho, inv_w = sw > 0.f ? 1.f / sw : (float)Wo;
    int oh_lo = (int)floorf(((float)ih - 1.f) * inv_h) - 2, oh_hi = (int)ceilf(((float)ih + 1.f) * inv_h) + 2;
    int ow_lo = (int)floorf(((float)iw - 1.f) * inv_w) - 2, ow_hi = (int)ceilf(((float)iw + 1.f) * inv_w) + 2;
    oh_lo = max(oh_lo, 0); oh_hi = min(oh_hi, Ho - 1);
    ow_lo = max(ow_lo, 0); ow_hi = min(ow_hi, Wo - 1);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    const T *base = gy + (int64_t)b * Ho * Wo * C + c * V;
    // the columns that sampled iw are the same for every row: found once (up to MAXM of them, in ascending order, so the
    // sum keeps the order of the plain double loop below, which stays as the path for larger scale factors)
    constexpr int MAXM = 6;
    int mcol[MAXM];
    float mw[MAXM];
    int nm = 0;
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        int x0, x1;
        float lx;
        src_index(ow, sw, align, Wi, x0, x1, lx);
        float wx = 0.f;
        if (x0 == iw) wx += 1.f - lx;
        if (x1 == iw) wx += lx;
        if (wx == 0.f) continue;
        if (nm < MAXM) {
#pragma unroll
            for (int k = 0; k < MAXM; ++k)
                if (k == nm) { mcol[k] = ow; mw[k] = wx; }
        }
        ++nm;
    }
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        int y0, y1;
        float ly;
        src_index(oh, sh, align, Hi, y0, y1, ly);
        float wy = 0.f;
        if (y0 == ih) wy += 1.f - ly;
        if (y1 == ih) wy += ly;
        if (wy == 0.f) continue;
        const T *row = base + (int64_t)oh * Wo * C;
        if (nm <= MAXM) {
#pragma unroll
            for (int k = 0; k < MAXM; ++k)
                if (k < nm) {
                    float g[V];
                    ldv<V>(row + (int64_t)mcol[k] * C, g);
                    const float wgt = wy * mw[k];
#pragma unroll
                    for (int q = 0; q < V; ++q) acc[q] += wgt * g[q];
                }
            continue;
        }
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
            int x0, x1;
            float lx;
            src_index(ow, sw, align, Wi, x0, x1, lx);
            float wx = 0.f;
            if (x0 == iw) wx += 1.f - lx;
            if (x1 == iw) wx += lx;
            if (wx == 0.f) continue;
            float g[V];
            ldv<V>(row + (int64_t)ow * C, g);
            const float wgt = wy * wx;
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] += wgt * g[q];
        }
    }
    stv<V>(gx + e * V, acc);
}

// ------------------------------------------------------------------------------------
// 3x3 stride-2 pad-1 max-pool (image stem, torchvision ResNet layout), NHWC.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_maxpool_fwd(const T *x, T *y, int B, int H, int W, int Ho, int Wo, int C4)
{
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    int64_t p = e / C4;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const int C = C4 * 4;
    float4 m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
    for (int dh = 0; dh < 3; ++dh) {
        const int ih = oh * 2 - 1 + dh;
        if (ih < 0 || ih >= H) continue;
        for (int dw = 0; dw < 3; ++dw) {
            const int iw = ow * 2 - 1 + dw;
            if (iw < 0 || iw >= W) continue;
            const float4 v = ld4(x + (((int64_t)b * H + ih) * W + iw) * C + c * 4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    st4(y + e * 4, m);
}

// forward that also records WHICH window position (dh*3+dw, first maximum in scan order) each output took: the backward
// is then a plain gather (4 index words + 4 gradient loads per input pixel, no re-scan of the windows, no divergence)
template <typename T>
__global__ void __launch_bounds__(256) k_maxpool_fwd_idx(const T *x, T *y, uint32_t *idx, int B, int H, int W, int Ho, int Wo, int C4)
{
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    int64_t p = e / C4;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const int C = C4 * 4;
    float m[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t pos[4] = {0, 0, 0, 0};
    for (int dh = 0; dh < 3; ++dh) {
        const int ih = oh * 2 - 1 + dh;
        if (ih < 0 || ih >= H) continue;
        for (int dw = 0; dw < 3; ++dw) {
            const int iw = ow * 2 - 1 + dw;
            if (iw < 0 || iw >= W) continue;
            const float4 v = ld4(x + (((int64_t)b * H + ih) * W + iw) * C + c * 4);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (vv[k] > m[k]) { m[k] = vv[k]; pos[k] = (uint32_t)(dh * 3 + dw); }
        }
    }
    st4(y + e * 4, make_float4(m[0], m[1], m[2], m[3]));
    idx[e] = pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24);
}

template <typename T>
__global__ void __launch_bounds__(256) k_maxpool_bwd_idx(const uint32_t *idx, const T *gy, T *gx, int B, int H, int W, int Ho, int Wo, int C4)
{
    const int64_t total = (int64_t)B * H * W * C4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    int64_t p = e / C4;
    const int iw = (int)(p % W); p /= W;
    const int ih = (int)(p % H);
    const int b = (int)(p / H);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oh = (ih + 1) / 2 - 1; oh <= (ih + 1) / 2; ++oh) {
        if (oh < 0 || oh >= Ho || ih < oh * 2 - 1 || ih > oh * 2 + 1) continue;
        for (int ow = (iw + 1) / 2 - 1; ow <= (iw + 1) / 2; ++ow) {
            if (ow < 0 || ow >= Wo || iw < ow * 2 - 1 || iw > ow * 2 + 1) continue;
            const int64_t o = (((int64_t)b * Ho + oh) * Wo + ow) * C4 + c;
            const uint32_t mine = (uint32_t)((ih - (oh * 2 - 1)) * 3 + (iw - (ow * 2 - 1)));
            const uint32_t w = idx[o];
            const float4 g = ld4(gy + o * 4);
            if ((w & 0xff) == mine) acc[0] += g.x;
            if (((w >> 8) & 0xff) == mine) acc[1] += g.y;
            if (((w >> 16) & 0xff) == mine) acc[2] += g.z;
            if ((w >> 24) == mine) acc[3] += g.w;
        }
    }
    st4(gx + e * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

// gather-form backward: input pixel collects gy from every window whose FIRST max (scan
// order dh,dw) is this pixel -- the same element ATen's max_pool2d backward routes to.
template <typename T>
__global__ void __launch_bounds__(256) k_maxpool_bwd(const T *x, const T *gy, T *gx, int B, int H, int W, int Ho, int Wo, int C4)
{
    const int64_t total = (int64_t)B * H * W * C4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    int64_t p = e / C4;
    const int iw = (int)(p % W); p /= W;
    const int ih = (int)(p % H);
    const int b = (int)(p / H);
    const int C = C4 * 4;
    const float4 me = ld4(x + e * 4);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float mev[4] = {me.x, me.y, me.z, me.w};
    for (int oh = (ih + 1) / 2 - 1; oh <= (ih + 1) / 2; ++oh) {
        if (oh < 0 || oh >= Ho || ih < oh * 2 - 1 || ih > oh * 2 + 1) continue;
        for (int ow = (iw + 1) / 2 - 1; ow <= (iw + 1) / 2; ++ow) {
            if (ow < 0 || ow >= Wo || iw < ow * 2 - 1 || iw > ow * 2 + 1) continue;
            // find the first arg-max of this window per channel
            float best[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
            int bpos[4] = {-1, -1, -1, -1};
            for (int dh = 0; dh < 3; ++dh) {
                const int hh = oh * 2 - 1 + dh;
                if (hh < 0 || hh >= H) continue;
                for (int dw = 0; dw < 3; ++dw) {
                    const int ww = ow * 2 - 1 + dw;
                    if (ww < 0 || ww >= W) continue;
                    const float4 v = ld4(x + (((int64_t)b * H + hh) * W + ww) * C + c * 4);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (vv[k] > best[k]) { best[k] = vv[k]; bpos[k] = hh * W + ww; }
                }
            }
            const float4 g = ld4(gy + (((int64_t)b * Ho + oh) * Wo + ow) * C + c * 4);
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (bpos[k] == ih * W + iw) acc[k] += gv[k];
        }
    }
    (void)mev;
    st4(gx + e * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

// Same routing with the pooled output y at hand: a pixel can only be a window's arg-max if it EQUALS the window's
// maximum, so a window costs one load of y (+ one of gy and a scan of the EARLIER window positions for ties) instead of
// nine loads of x for every (pixel, window) pair.
template <typename T>
__global__ void __launch_bounds__(256) k_maxpool_bwd_y(const T *x, const T *y, const T *gy, T *gx, int B, int H, int W, int Ho, int Wo, int C4)
{
    const int64_t total = (int64_t)B * H * W * C4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    int64_t p = e / C4;
    const int iw = (int)(p % W); p /= W;
    const int ih = (int)(p % H);
    const int b = (int)(p / H);
    const int C = C4 * 4;
    const float4 me = ld4(x + e * 4);
    const float mev[4] = {me.x, me.y, me.z, me.w};
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oh = (ih + 1) / 2 - 1; oh <= (ih + 1) / 2; ++oh) {
        if (oh < 0 || oh >= Ho || ih < oh * 2 - 1 || ih > oh * 2 + 1) continue;
        for (int ow = (iw + 1) / 2 - 1; ow <= (iw + 1) / 2; ++ow) {
            if (ow < 0 || ow >= Wo || iw < ow * 2 - 1 || iw > ow * 2 + 1) continue;
            const int64_t o = (((int64_t)b * Ho + oh) * Wo + ow) * C + c * 4;
            const float4 m = ld4(y + o);
            const float mv[4] = {m.x, m.y, m.z, m.w};
            bool cand[4];
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) { cand[k] = mev[k] == mv[k]; any |= cand[k]; }
            if (!any) continue;
            // ties: an earlier position (scan order dh, dw) holding the same maximum wins instead
            const int mydh = ih - (oh * 2 - 1), mydw = iw - (ow * 2 - 1);
            for (int dh = 0; dh <= mydh; ++dh) {
                const int hh = oh * 2 - 1 + dh;
                if (hh < 0) continue;
                const int dwend = dh == mydh ? mydw : 3;
                for (int dw = 0; dw < dwend; ++dw) {
                    const int ww = ow * 2 - 1 + dw;
                    if (ww < 0 || ww >= W) continue;
                    const float4 v = ld4(x + (((int64_t)b * H + hh) * W + ww) * C + c * 4);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) cand[k] = cand[k] && !(vv[k] == mv[k]);
                }
                if (!(cand[0] | cand[1] | cand[2] | cand[3])) break;     // e.g. an all-zero (post-ReLU) window: the first position wins
            }
            const float4 g = ld4(gy + o);
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cand[k]) acc[k] += gv[k];
        }
    }
    st4(gx + e * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
}


// ------------------------------------------------------------------------------------
// Detection heads.  head [B,h,w,Cp]: channels 0..3 class logits, 4..17 box offsets.
// pred [B,32,h,w] fp32 = cat(softmax(l0,l1), softmax(l2,l3), reg14, decode(reg14)).
// model.py:168-172 (softmax pairs), :126-136 (decode), :204 (concat).  fp32 math.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_head_fwd(const T *head, int Cp, const float *anc, float *pred, int B, int hw)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)B * hw) return;
    const int b = (int)(p / hw), q = (int)(p - (int64_t)b * hw);
    const T *hp = head + p * Cp;
    float v[20];
#pragma unroll
    for (int k = 0; k < 20; k += 4) {
        const float4 t = ld4(hp + k);
        v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w;
    }
    float *o = pred + (int64_t)b * 32 * hw + q;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float l0 = v[2 * a], l1 = v[2 * a + 1];
        const float m = fmaxf(l0, l1);
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float inv = 1.0f / (e0 + e1);
        o[(int64_t)(2 * a) * hw] = e0 * inv;
        o[(int64_t)(2 * a + 1) * hw] = e1 * inv;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float *r = v + 4 + 7 * a;
        float an[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) an[k] = anc[(int64_t)(7 * a + k) * hw + q];
        const float diag = sqrtf(an[3] * an[3] + an[4] * an[4]);
        float box[7];
        box[0] = r[0] * diag + an[0];
        box[1] = r[1] * diag + an[1];
        box[2] = r[2] * an[5] + an[2];
        box[3] = expf(r[3]) * an[3];
        box[4] = expf(r[4]) * an[4];
        box[5] = expf(r[5]) * an[5];
        const float t = r[6] + an[6];
        box[6] = atan2f(sinf(t), cosf(t));
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            o[(int64_t)(4 + 7 * a + k) * hw] = r[k];
            o[(int64_t)(18 + 7 * a + k) * hw] = box[k];
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_head_bwd(int Cp, const float *anc, const float *pred, const float *gpred, T *ghead, int B, int hw)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (int64_t)B * hw) return;
    const int b = (int)(p / hw), q = (int)(p - (int64_t)b * hw);
    const float *o = pred + (int64_t)b * 32 * hw + q;
    const float *g = gpred + (int64_t)b * 32 * hw + q;
    float out[20];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float p0 = o[(int64_t)(2 * a) * hw], p1 = o[(int64_t)(2 * a + 1) * hw];
        const float g0 = g[(int64_t)(2 * a) * hw], g1 = g[(int64_t)(2 * a + 1) * hw];
        const float dot = g0 * p0 + g1 * p1;
        out[2 * a] = p0 * (g0 - dot);
        out[2 * a + 1] = p1 * (g1 - dot);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float al = anc[(int64_t)(7 * a + 3) * hw + q], aw = anc[(int64_t)(7 * a + 4) * hw + q], ah = anc[(int64_t)(7 * a + 5) * hw + q];
        const float diag = sqrtf(al * al + aw * aw);
        float dj[7];
        dj[0] = diag; dj[1] = diag; dj[2] = ah;
        dj[3] = o[(int64_t)(18 + 7 * a + 3) * hw];  // d exp(r)*anc / dr = box value
        dj[4] = o[(int64_t)(18 + 7 * a + 4) * hw];
        dj[5] = o[(int64_t)(18 + 7 * a + 5) * hw];
        dj[6] = 1.0f;                               // d atan2(sin t, cos t)/dt
#pragma unroll
        for (int k = 0; k < 7; ++k)
            out[4 + 7 * a + k] = g[(int64_t)(4 + 7 * a + k) * hw] + g[(int64_t)(18 + 7 * a + k) * hw] * dj[k];
    }
    out[18] = 0.f; out[19] = 0.f;
    T *hp = ghead + p * Cp;
#pragma unroll
    for (int k = 0; k < 20; k += 4) st4(hp + k, make_float4(out[k], out[k + 1], out[k + 2], out[k + 3]));
    for (int k = 20; k < Cp; k += 4) st4(hp + k, make_float4(0.f, 0.f, 0.f, 0.f));
}

// ------------------------------------------------------------------------------------
// Per-step weight preparation, one launch for every convolution (blockIdx.y = conv).
// Folds the eval-mode BatchNorm (model.py:20,24,29; eval per SURVEY.md F4) into the weights:
//   scale = gamma*rsqrt(var+eps), shift = beta - mean*scale,  w' = cast(scale*W).
// ------------------------------------------------------------------------------------
#define DCF_PREP_MAXCONV 256
template <typename T>
__global__ void __launch_bounds__(256) k_weight_prep(const dcf_conv_param *table, int nconv, const float *params, const float *buffers, char *warena,
                                                     float *ssarena, float eps)
{
    // 64(co) x 64(ci) tiles per tap, 16 B per lane: whole 256-B rows of the fp32 master weights in, whole 128-B (bf16) rows
    // of both images out; the dgrad image ([ci][tap][co]) goes through an LDS transpose (pitch 65: column reads conflict free).
    __shared__ float tile[64][65];
    // The (conv, tap, 64 x 64 tile) list of all layers is dealt round-robin over the grid: a layer gets workgroups in proportion
    // to its size (a fixed number of workgroups per layer left the 2.4 M-weight layers to 96 workgroups while 95 of the 96 of a
    // one-tile layer exited at once).  Every workgroup builds the prefix sums of the tiles per layer itself (<= 256 layers).
    __shared__ int pre[DCF_PREP_MAXCONV + 1];
    for (int i = threadIdx.x; i < nconv; i += blockDim.x) {
        const dcf_conv_param &e = table[i];
        pre[i + 1] = e.taps * ((e.cout_pad + 63) / 64) * ((e.cin + 63) / 64);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        pre[0] = 0;
        for (int i = 0; i < nconv; ++i) pre[i + 1] += pre[i];
    }
    __syncthreads();
    const int total = pre[nconv];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 lanes x 4 elements per row, 16 rows per pass
    for (int gt = blockIdx.x; gt < total; gt += gridDim.x) {
        int lo = 0, hi = nconv;                                // last layer with pre[layer] <= gt
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= gt) lo = mid; else hi = mid; }
        const dcf_conv_param d = table[lo];
        const int t = gt - pre[lo];
        const int K = d.taps * d.cin;
        const int cit = (d.cin + 63) / 64;
        T *wf = reinterpret_cast<T *>(warena + d.wfwd_off);
        T *wd = d.wdgrad_off >= 0 ? reinterpret_cast<T *>(warena + d.wdgrad_off) : nullptr;
        const int tap = t % d.taps;
        const int r = t / d.taps;
        const int c0 = (r % cit) * 64, o0 = (r / cit) * 64;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = o0 + ty + 16 * k, ci = c0 + tx * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool in = (co < d.cout_pad) && (ci < d.cin);
            if (in && co < d.cout) {
                float scale = 1.f;
                if (d.gamma_off >= 0) scale = params[d.gamma_off + co] * rsqrtf(buffers[d.var_off + co] + eps);
                v = ld4(params + d.w_off + (int64_t)co * K + tap * d.cin + ci);
                v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            }
            if (in) st4(wf + (int64_t)co * K + tap * d.cin + ci, v);
            float *row = &tile[ty + 16 * k][tx * 4];
            row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
        }
        __syncthreads();
        if (wd) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = c0 + ty + 16 * k, co = o0 + tx * 4;
                if (ci < d.cin && co < d.cout_pad) {
                    const int lr = ty + 16 * k;
                    st4(wd + ((int64_t)ci * d.taps + tap) * d.cout_pad + co,
                        make_float4(tile[tx * 4][lr], tile[tx * 4 + 1][lr], tile[tx * 4 + 2][lr], tile[tx * 4 + 3][lr]));
                }
            }
        }
        __syncthreads();
        // scale | shift (one thread per output channel, by the workgroup that has the layer's first tile)
        if (t != 0) continue;
        for (int co = threadIdx.x; co < d.cout_pad; co += blockDim.x) {
            float scale = 0.f, shift = 0.f;
            if (co < d.cout) {
                scale = 1.f;
                if (d.gamma_off >= 0) {
                    scale = params[d.gamma_off + co] * rsqrtf(buffers[d.var_off + co] + eps);
                    shift = params[d.beta_off + co] - buffers[d.mean_off + co] * scale;
                }
            }
            ssarena[d.shift_off + co] = scale;
            ssarena[d.shift_off + d.cout_pad + co] = shift;
        }
    }
}

// Gradient finalisation, one block per (conv, output channel): fixed-order reduction of the
// wgrad slabs (16 B per lane, slab order, 8 loads in flight per lane), then the folded-BN chain rule
//   dW = scale*G ; dbeta = sum g ; dgamma = (<W,G> - mean*dbeta) * rsqrt(var+eps).
// rows = kernel-argument copy of the layers' first rows (prefix sums of cout): the grid is exactly one block per (conv, output
// channel) instead of max_cout x nconv blocks of which three quarters exit at once
#define DCF_FIN_MAXCONV 256
struct FinRows { int n; int first[DCF_FIN_MAXCONV + 1]; };

__global__ void __launch_bounds__(256) k_wgrad_finalize(const dcf_conv_param *table, const float *params, const float *buffers,
                                                        const float *ssarena, const float *slabs, const float *gsum, float *grads, float eps,
                                                        int grouped, FinRows rows)
{
    int li, co;
    if (rows.n > 0) {                            // binary search of the block's layer in the (scalar, cached) kernel arguments
        int lo = 0, hi = rows.n;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((int)blockIdx.x >= rows.first[mid]) lo = mid; else hi = mid;
        }
        li = lo; co = blockIdx.x - rows.first[lo];
    } else {
        li = blockIdx.y; co = blockIdx.x;
    }
    const dcf_conv_param d = table[li];
    if (co >= d.cout) return;
    const int K = d.taps * d.cin;
    const int64_t slab_elems = (int64_t)d.cout_pad * K;
    const float scale = ssarena[d.shift_off + co];
    float dot = 0.f;
    // dbeta: per-wave partial sums written by the wgrad kernel, reduced with a fixed thread mapping (loaded first: independent of
    // the slab reduction below, so their latency hides under it)
    float db = 0.f;
    if (d.gamma_off >= 0)
        for (int sp = threadIdx.x; sp < 4 * d.nsplit; sp += blockDim.x) db += gsum[d.gsum_off + (int64_t)sp * d.cout_pad + co];
    // Rows shorter than the block (K/4 < 256 lanes: the 1x1 and 32-channel layers, which are also the ones with hundreds
    // of slabs) are reduced by G = 256/(K/4) thread groups, group g taking slabs g, g+G, ...; the group sums meet in LDS
    // and are added in group order, so the result does not depend on timing.
    __shared__ float4 gpart[256];
    const int K4 = K >> 2;
    const int G = (K4 >= 256 || !grouped) ? 1 : 256 / K4;
    const float sc = d.gamma_off >= 0 ? scale : 1.f;
    // one finished column: stem zeros, <W, G> for dgamma, scaled store
    auto finish = [&](int k, float4 g4) {
        const int64_t e = (int64_t)co * K + k;
        const float4 w = ld4(params + d.w_off + e);
        // stem weights are stored [Cout][7][8][4]: tap kw=7 and channel 3 are structural zeros
        if (d.flags & 1) {
            if (((k & 31) >> 2) == 7) g4.x = g4.y = g4.z = 0.f;
            g4.w = 0.f;
        }
        dot += (w.x * g4.x + w.y * g4.y) + (w.z * g4.z + w.w * g4.w);
        st4(grads + d.w_off + e, make_float4(sc * g4.x, sc * g4.y, sc * g4.z, sc * g4.w));
    };
    if (G > 1) {
        // short rows: group grp sums the slabs grp, grp + G, ... (8 loads in flight), the group sums are added in group order
        const int grp = threadIdx.x / K4;
        const int k = (threadIdx.x - grp * K4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (grp < G) {
            const float *sp = slabs + d.slab_off + (int64_t)co * K + k;
            int sidx = grp;
            for (; sidx + 7 * G < d.nsplit; sidx += 8 * G) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ld4(sp + (int64_t)(sidx + u * G) * slab_elems);
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; sidx < d.nsplit; sidx += G) {
                const float4 v = ld4(sp + (int64_t)sidx * slab_elems);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            gpart[threadIdx.x] = acc;
        }
        __syncthreads();
        if (grp == 0) {
            for (int g2 = 1; g2 < G; ++g2) {
                const float4 v = gpart[g2 * K4 + threadIdx.x];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            finish(k, acc);
        }
    } else {
        // long rows: a thread owns the columns tid, tid + 256, ... and walks the slabs of up to FOUR of them together, in
        // slab order, with 8 loads in flight (a row of 288 or 576 vectors is then one or two dependent chains, not two or three)
        for (int kb = 0; kb < K4; kb += 1024) {
            const int ncol = min(4, (K4 - kb + 255) >> 8);
            float4 acc[4];
            const float *sp[4];
            bool on[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int k4 = kb + c * 256 + (int)threadIdx.x;
                on[c] = k4 < K4;
                sp[c] = slabs + d.slab_off + (int64_t)co * K + (on[c] ? k4 : 0) * 4;
            }
            if (ncol == 1) {
                if (on[0]) {
                    int sidx = 0;
                    for (; sidx + 7 < d.nsplit; sidx += 8) {
                        float4 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = ld4(sp[0] + (int64_t)(sidx + u) * slab_elems);
#pragma unroll
                        for (int u = 0; u < 8; ++u) { acc[0].x += v[u].x; acc[0].y += v[u].y; acc[0].z += v[u].z; acc[0].w += v[u].w; }
                    }
                    for (; sidx < d.nsplit; ++sidx) {
                        const float4 v = ld4(sp[0] + (int64_t)sidx * slab_elems);
                        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
                    }
                }
            } else if (ncol == 2) {
                int sidx = 0;
                for (; sidx + 3 < d.nsplit; sidx += 4) {
                    float4 v[2][4];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[c][u] = on[c] ? ld4(sp[c] + (int64_t)(sidx + u) * slab_elems) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int u = 0; u < 4; ++u) { acc[c].x += v[c][u].x; acc[c].y += v[c][u].y; acc[c].z += v[c][u].z; acc[c].w += v[c][u].w; }
                }
                for (; sidx < d.nsplit; ++sidx)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        if (on[c]) {
                            const float4 v = ld4(sp[c] + (int64_t)sidx * slab_elems);
                            acc[c].x += v.x; acc[c].y += v.y; acc[c].z += v.z; acc[c].w += v.w;
                        }
            } else {
                int sidx = 0;
                for (; sidx + 1 < d.nsplit; sidx += 2) {
                    float4 v[4][2];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) v[c][u] = on[c] ? ld4(sp[c] + (int64_t)(sidx + u) * slab_elems) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int u = 0; u < 2; ++u) { acc[c].x += v[c][u].x; acc[c].y += v[c][u].y; acc[c].z += v[c][u].z; acc[c].w += v[c][u].w; }
                }
                for (; sidx < d.nsplit; ++sidx)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (on[c]) {
                            const float4 v = ld4(sp[c] + (int64_t)sidx * slab_elems);
                            acc[c].x += v.x; acc[c].y += v.y; acc[c].z += v.z; acc[c].w += v.w;
                        }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (on[c]) finish((kb + c * 256 + (int)threadIdx.x) * 4, acc[c]);
        }
    }
    if (d.gamma_off < 0) return;
    __shared__ float red[8];
    dot = wave_sum(dot);
    db = wave_sum(db);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = dot; red[4 + (threadIdx.x >> 6)] = db; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        const float dbeta = (red[4] + red[5]) + (red[6] + red[7]);
        const float invstd = rsqrtf(buffers[d.var_off + co] + eps);
        grads[d.beta_off + co] = dbeta;
        grads[d.gamma_off + co] = (tot - buffers[d.mean_off + co] * dbeta) * invstd;
    }
}

// ------------------------------------------------------------------------------------
// Fused Adam (train.py:28,36), torch.optim.Adam semantics, flat arena, float4 per lane.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_adam(float *p, const float *g, float *m, float *v, int64_t n, float lr_over_bc1, float b1, float b2,
                                              float inv_sqrt_bc2, float eps, float gscale)
{
    const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n) {
        float4 pp = ld4(p + i4), gg = ld4(g + i4), mm = ld4(m + i4), vv = ld4(v + i4);
        float *P = &pp.x, *G = &gg.x, *M = &mm.x, *V = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = G[k] * gscale;
            M[k] = b1 * M[k] + (1.f - b1) * gk;
            V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
            P[k] -= lr_over_bc1 * M[k] / (sqrtf(V[k]) * inv_sqrt_bc2 + eps);
        }
        st4(p + i4, pp); st4(m + i4, mm); st4(v + i4, vv);
    } else {
        for (int64_t i = i4; i < n; ++i) {
            const float gk = g[i] * gscale;
            m[i] = b1 * m[i] + (1.f - b1) * gk;
            v[i] = b2 * v[i] + (1.f - b2) * gk * gk;
            p[i] -= lr_over_bc1 * m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps);
        }
    }
}

template <typename TS, typename TD>
__global__ void __launch_bounds__(256) k_cast(const TS *s, TD *d, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) st4(d + i * 4, ld4(s + i * 4));
}

// y[p][c] += cnt[p]*b2[c]  /  gb2[c] += sum_p cnt[p]*gy[p][c]   (fc2 bias of the fusion K-sum)
template <typename T>
__global__ void __launch_bounds__(256) k_rowscale_bias_fwd(T *y, const float *cnt, const float *b2, int64_t nvec, int C4)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nvec) return;
    const int64_t p = e / C4;
    const int c = (int)(e - p * C4) * 4;
    const float k = cnt[p];
    float4 v = ld4(y + e * 4);
    const float4 b = *reinterpret_cast<const float4 *>(b2 + c);
    v.x += k * b.x; v.y += k * b.y; v.z += k * b.z; v.w += k * b.w;
    st4(y + e * 4, v);
}

// V channels per thread and four independent loads in flight per thread: the kernel is a latency-bound stream (a grid that fits
// the atomics at its end -- 256 workgroups -- has to keep ~8 MB in flight to run at the memory's rate)
template <typename T, int V>
__global__ void __launch_bounds__(256) k_rowscale_bias_bwd(const T *gy, const float *cnt, float *gb2, int64_t nvec, int cgroups, int64_t stride)
{
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < cgroups * V; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < stride) {
        const int cg = (int)(t % cgroups);
        float acc[V];
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] = 0.f;
        int64_t p = t / cgroups;                          // pixel of element e; stride is a multiple of cgroups
        const int64_t pstep = stride / cgroups;
        int64_t e = t;
        for (; e + 3 * stride < nvec; e += 4 * stride, p += 4 * pstep) {
            float g[4][V], k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ldv<V>(gy + (e + u * stride) * V, g[u]);
                k[u] = cnt[p + u * pstep];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < V; ++q) acc[q] += k[u] * g[u][q];
        }
        for (; e < nvec; e += stride, p += pstep) {
            float g[V];
            ldv<V>(gy + e * V, g);
            const float k = cnt[p];
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] += k * g[q];
        }
#pragma unroll
        for (int q = 0; q < V; ++q) atomicAdd(&sm[cg * V + q], acc[q]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cgroups * V; i += blockDim.x) atomicAdd(&gb2[i], sm[i]);
}

// At a fusion site (/root/reference/model.py:199-203 is the TODO) the gradient of the site's output feeds the
// fusion branch unmasked (fc2's bias gradient is its cnt-weighted channel sum) and the stage's last block masked by that block's
// ReLU.  One pass instead of two (k_rowscale_bias_bwd, then k_relu_bwd_chansum in place): reads gy, y and cnt, writes the masked
// gradient to its OWN tensor (the fusion backward still reads gy), accumulates gb2[c] += sum_p cnt[p] * gy[p][c].
template <typename T, int V>
__global__ void __launch_bounds__(256) k_relu_mask_rowscale_bwd(const T *__restrict__ gy, const T *__restrict__ y, const float *__restrict__ cnt,
                                                                T *__restrict__ gout, float *gb2, int64_t nvec, int cgroups, int64_t stride)
{
    // A latency-bound stream like k_rowscale_bias_bwd: 256 workgroups (the C same-address atomics per workgroup at the end stay
    // cheap), so every thread keeps FOUR elements' loads in flight (8 x 16 B + 4 counts); the workgroup's channel sums are formed
    // without atomics (every thread parks its V sums in LDS, one thread per channel adds its column in thread order).
    extern __shared__ float sm[];  // [256][V]
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc[V];
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
    if (t < stride) {
        int64_t p = t / cgroups;                          // pixel of element e; stride is a multiple of cgroups
        const int64_t pstep = stride / cgroups;
        int64_t e = t;
        for (; e + 3 * stride < nvec; e += 4 * stride, p += 4 * pstep) {
            float g[4][V], yy[4][V], k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ldv<V>(gy + (e + u * stride) * V, g[u]);
                ldv<V>(y + (e + u * stride) * V, yy[u]);
                k[u] = cnt[p + u * pstep];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float m[V];
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    acc[q] += k[u] * g[u][q];
                    m[q] = yy[u][q] > 0.f ? g[u][q] : 0.f;
                }
                stv<V>(gout + (e + u * stride) * V, m);
            }
        }
        for (; e < nvec; e += stride, p += pstep) {
            float g[V], yy[V], m[V];
            ldv<V>(gy + e * V, g);
            ldv<V>(y + e * V, yy);
            const float k = cnt[p];
#pragma unroll
            for (int q = 0; q < V; ++q) {
                acc[q] += k * g[q];
                m[q] = yy[q] > 0.f ? g[q] : 0.f;
            }
            stv<V>(gout + e * V, m);
        }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) sm[threadIdx.x * V + q] = acc[q];
    __syncthreads();
    // channel i = group j x lane q of it; the threads of this workgroup whose elements belong to group j: t % cgroups == j
    for (int i = threadIdx.x; i < cgroups * V; i += blockDim.x) {
        const int j = i / V, q = i - j * V;
        const int base = (int)(((int64_t)blockIdx.x * blockDim.x) % cgroups);
        int first = j - base;
        if (first < 0) first += cgroups;
        float sum = 0.f;
        for (int th = first; th < (int)blockDim.x; th += cgroups) sum += sm[th * V + q];
        atomicAdd(&gb2[i], sum);
    }
}

// ------------------------------------------------------------------------------------
// Train-mode BatchNorm2d (model.py:20,24,29 when the module is in .train(): batch statistics).
//   stats : per-channel sum / sum-of-squares partials per workgroup, finalised in double
//           (mean, biased var -> invstd; running stats updated with momentum, unbiased var)
//   apply : y = act(gamma*(x-mean)*invstd + beta + res)
//   bwd   : dbeta = sum g, dgamma = sum g*xhat, dx = gamma*invstd*(g - dbeta/M - xhat*dgamma/M)
// ------------------------------------------------------------------------------------
// V channels per thread (8 = 16-byte accesses for the 16-bit types when the channel count allows, else 4); a thread keeps its
// channel group for its whole grid-stride walk and has FOUR elements' loads in flight per trip.  The workgroup's sums are formed
// without atomics: every thread parks its 2 V sums in LDS and each of the 2 C outputs is summed, in thread order, by one thread
// (round 5: LDS atomics from 256 threads onto C / V channel groups serialised 8- to 64-fold -- ~11 us per launch whatever the
// tensor's size, 1.3 ms of the 8.7 ms train-mode step over its 62 + 62 launches; the sums are now bitwise reproducible too).
template <typename T, bool BWD, int V>
__global__ void __launch_bounds__(256) k_bn_partial(const T *x, const T *g, const float *mean, const float *invstd, float *partial,
                                                    int64_t nvec, int cgroups, int64_t stride)
{
    constexpr int PITCH = 2 * V + 1;               // odd pitch: a column walk touches every bank once
    __shared__ float buf[256 * PITCH];
    const int C = cgroups * V;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float a[V], b[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { a[k] = 0.f; b[k] = 0.f; }
    if (t < stride) {
        const int cg = (int)(t % cgroups);
        float mu[V], is[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { mu[k] = 0.f; is[k] = 1.f; }
        if (BWD) {
#pragma unroll
            for (int k = 0; k < V; ++k) { mu[k] = mean[cg * V + k]; is[k] = invstd[cg * V + k]; }
        }
        auto add = [&](const float (&xx)[V], const float (&gg)[V]) __attribute__((always_inline)) {
            if (BWD) {
#pragma unroll
                for (int k = 0; k < V; ++k) { a[k] += gg[k]; b[k] += gg[k] * ((xx[k] - mu[k]) * is[k]); }
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) { a[k] += xx[k]; b[k] += xx[k] * xx[k]; }
            }
        };
        int64_t e = t;
        for (; e + 3 * stride < nvec; e += 4 * stride) {
            float x0[V], x1[V], x2[V], x3[V], g0[V], g1[V], g2[V], g3[V];
            ldv<V>(x + e * V, x0); ldv<V>(x + (e + stride) * V, x1); ldv<V>(x + (e + 2 * stride) * V, x2); ldv<V>(x + (e + 3 * stride) * V, x3);
            if (BWD) { ldv<V>(g + e * V, g0); ldv<V>(g + (e + stride) * V, g1); ldv<V>(g + (e + 2 * stride) * V, g2); ldv<V>(g + (e + 3 * stride) * V, g3); }
            add(x0, g0); add(x1, g1); add(x2, g2); add(x3, g3);
        }
        for (; e < nvec; e += stride) {
            float x0[V], g0[V];
            ldv<V>(x + e * V, x0);
            if (BWD) ldv<V>(g + e * V, g0);
            add(x0, g0);
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) { buf[threadIdx.x * PITCH + k] = a[k]; buf[threadIdx.x * PITCH + V + k] = b[k]; }
    __syncthreads();
    // output i = (which sum, channel c): the threads of channel group c / V are tid0, tid0 + cgroups, ... (a thread's group is
    // (first thread's group + tid) mod cgroups; threads past `stride` parked zeros)
    const int g0 = (int)(((int64_t)blockIdx.x * blockDim.x) % cgroups);
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        const int which = i >= C, c = i - which * C;
        const int cg = c / V, k = c - cg * V;
        int tid = cg - g0;
        tid += tid < 0 ? cgroups : 0;
        float sum = 0.f;
        for (; tid < 256; tid += cgroups) sum += buf[tid * PITCH + which * V + k];
        partial[(int64_t)blockIdx.x * 2 * C + i] = sum;
    }
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wave per channel: lanes stride over the workgroup partials (fixed mapping => reproducible), double sums
__global__ void __launch_bounds__(64) k_bn_stats_final(const float *partial, int nblk, int C, double count, float eps, float momentum,
                                                       float *mean, float *invstd, float *running_mean, float *running_var)
{
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) { s += partial[(int64_t)b * 2 * C + c]; q += partial[(int64_t)b * 2 * C + C + c]; }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (threadIdx.x != 0) return;
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

__global__ void __launch_bounds__(64) k_bn_bwd_final(const float *partial, int nblk, int C, float *dgamma, float *dbeta)
{
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) { s += partial[(int64_t)b * 2 * C + c]; q += partial[(int64_t)b * 2 * C + C + c]; }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (threadIdx.x == 0) { dbeta[c] = (float)s; dgamma[c] = (float)q; }
}

template <typename T, int V>
__global__ void __launch_bounds__(256) k_bn_apply_fwd(const T *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                                                      const T *res, T *y, int64_t nvec, int cgroups, int relu)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nvec) return;
    const int c = (int)(e % cgroups) * V;
    float xx[V], o[V];
    ldv<V>(x + e * V, xx);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = (xx[k] - mean[c + k]) * invstd[c + k] * gamma[c + k] + beta[c + k];
    if (res) {
        float r[V];
        ldv<V>(res + e * V, r);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] += r[k];
    }
    if (relu) {
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = fmaxf(o[k], 0.f);
    }
    stv<V>(y + e * V, o);
}

template <typename T, int V>
__global__ void __launch_bounds__(256) k_bn_apply_bwd(const T *g, const T *x, const float *mean, const float *invstd, const float *gamma,
                                                      const float *dgamma, const float *dbeta, T *dx, int64_t nvec, int cgroups, float inv_count)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nvec) return;
    const int c = (int)(e % cgroups) * V;
    float xx[V], gg[V], o[V];
    ldv<V>(x + e * V, xx);
    ldv<V>(g + e * V, gg);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float xh = (xx[k] - mean[c + k]) * invstd[c + k];
        o[k] = gamma[c + k] * invstd[c + k] * (gg[k] - dbeta[c + k] * inv_count - xh * dgamma[c + k] * inv_count);
    }
    stv<V>(dx + e * V, o);
}

}  // namespace

// ================================================================== C ABI
extern "C" int dcf_nchw_to_nhwc(int dtype, const float *x, void *y, int B, int C, int H, int W, dcf_stream_t stream)
{
    DCF_REQUIRE(x && y && C % 4 == 0, "dcf_nchw_to_nhwc: C must be a multiple of 4");
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, {
        if (C % 8 == 0) DCF_LAUNCH_B("nchw_to_nhwc", (double)total * C * (4.0 + sizeof(T)), s, hipLaunchKernelGGL((k_nchw_to_nhwc<T, 8>), dim3(cdiv(total, 256)), dim3(256), 0, s, x, (T *)y, C, HW, total));
        else DCF_LAUNCH_B("nchw_to_nhwc", (double)total * C * (4.0 + sizeof(T)), s, hipLaunchKernelGGL((k_nchw_to_nhwc<T, 4>), dim3(cdiv(total, 256)), dim3(256), 0, s, x, (T *)y, C, HW, total));
    })
    return DCF_OK;
}

extern "C" int dcf_nhwc_to_nchw(int dtype, const void *x, float *y, int B, int C, int H, int W, dcf_stream_t stream)
{
    DCF_REQUIRE(x && y && C % 4 == 0, "dcf_nhwc_to_nchw: C must be a multiple of 4");
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("nhwc_to_nchw", (double)total * C * (4.0 + sizeof(T)), s, hipLaunchKernelGGL(k_nhwc_to_nchw<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const T *)x, y, C, HW, total)); })
    return DCF_OK;
}

extern "C" int dcf_image_to_nhwc4(int dtype, const uint8_t *img, void *y, int B, int H, int W, dcf_stream_t stream)
{
    DCF_REQUIRE(img && y, "dcf_image_to_nhwc4: null pointer");
    const int64_t total = (int64_t)B * (H + 6) * (W + 8);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("image_to_nhwc4", (double)B * H * W * 3 + (double)total * 4 * sizeof(T), s, hipLaunchKernelGGL(k_image_to_nhwc4<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, img, (T *)y, B, H, W)); })
    return DCF_OK;
}

static inline int64_t chan_stride(int64_t nvec, int cgroups, int &blocks, int kthreads = 256)
{
    static DcfOpt env_o("CHANSUM_KTHREADS"); const char *env = env_o.str();
    const int64_t cap = (env ? atoi(env) : kthreads) * 1024ll;
    int64_t want = nvec < cap ? nvec : cap;                // 256 k threads = ~1024 blocks of 256 threads
    if (want < cgroups) want = cgroups;
    const int64_t stride = want / cgroups * cgroups;
    blocks = cdiv(stride, 256);
    return stride;
}

extern "C" int dcf_relu_bwd_chansum(int dtype, void *gy, const void *y, float *gsum, int64_t npix, int C, int relu,
                                    dcf_stream_t stream)
{
    DCF_REQUIRE(gy && C % 4 == 0 && (!relu || y), "dcf_relu_bwd_chansum: bad arguments");
    const int V = C % 8 == 0 ? 8 : 4;
    const int cg = C / V;
    const int64_t nvec = npix * cg;
    if (nvec == 0) return DCF_OK;
    int blocks;
    const int64_t stride = chan_stride(nvec, cg, blocks);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, {
        const double bytes = (double)npix * C * sizeof(T) * (relu ? 3 : 1);
        if (V == 8) DCF_LAUNCH_B("relu_bwd_chansum", bytes, s, hipLaunchKernelGGL((k_relu_bwd_chansum<T, 8>), dim3(blocks), dim3(256), sizeof(float) * C, s, (T *)gy, (const T *)y, gsum, nvec, cg, relu, stride));
        else DCF_LAUNCH_B("relu_bwd_chansum", bytes, s, hipLaunchKernelGGL((k_relu_bwd_chansum<T, 4>), dim3(blocks), dim3(256), sizeof(float) * C, s, (T *)gy, (const T *)y, gsum, nvec, cg, relu, stride));
    })
    return DCF_OK;
}

extern "C" int dcf_resize_bilinear_fwd(int dtype, const void *x, const void *add, void *y, int B, int Hi, int Wi, int Ho, int Wo,
                                       int C, int align_corners, dcf_stream_t stream)
{
    DCF_REQUIRE(x && y && C % 4 == 0, "dcf_resize_bilinear_fwd: bad arguments");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    const float sh = resize_scale(Hi, Ho, align_corners), sw = resize_scale(Wi, Wo, align_corners);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, {
        DCF_LAUNCH_B("resize_fwd", ((double)B * Hi * Wi * C + (double)total * 4 * (add ? 2 : 1)) * sizeof(T), s, hipLaunchKernelGGL(k_resize_fwd<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const T *)x, (const T *)add, (T *)y, B, Hi, Wi,
                                                        Ho, Wo, C / 4, align_corners, sh, sw));
    })
    return DCF_OK;
}

extern "C" int dcf_resize_bilinear_bwd(int dtype, const void *gy, void *gx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                       int align_corners, dcf_stream_t stream)
{
    DCF_REQUIRE(gy && gx && C % 4 == 0, "dcf_resize_bilinear_bwd: bad arguments");
    const int V = C % 8 == 0 ? 8 : 4;
    const int64_t total = (int64_t)B * Hi * Wi * (C / V);
    const float sh = resize_scale(Hi, Ho, align_corners), sw = resize_scale(Wi, Wo, align_corners);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, {
        const double bytes = ((double)B * Ho * Wo * C + (double)B * Hi * Wi * C) * sizeof(T);
        if (V == 8) DCF_LAUNCH_B("resize_bwd", bytes, s, hipLaunchKernelGGL((k_resize_bwd<T, 8>), dim3(cdiv(total, 256)), dim3(256), 0, s, (const T *)gy, (T *)gx, B, Hi, Wi, Ho, Wo, C / 8,
                                                        align_corners, sh, sw));
        else DCF_LAUNCH_B("resize_bwd", bytes, s, hipLaunchKernelGGL((k_resize_bwd<T, 4>), dim3(cdiv(total, 256)), dim3(256), 0, s, (const T *)gy, (T *)gx, B, Hi, Wi, Ho, Wo, C / 4,
                                                        align_corners, sh, sw));
    })
    return DCF_OK;
}

extern "C" int dcf_maxpool3x3s2_fwd(int dtype, const void *x, void *y, int B, int H, int W, int Ho, int Wo, int C,
                                    dcf_stream_t stream)
{
    DCF_REQUIRE(x && y && C % 4 == 0 && Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "dcf_maxpool3x3s2_fwd: bad arguments");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("maxpool_fwd", ((double)B * H * W * C + (double)total * 4) * sizeof(T), s, hipLaunchKernelGGL(k_maxpool_fwd<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const T *)x, (T *)y, B, H, W, Ho, Wo, C / 4)); })
    return DCF_OK;
}

extern "C" int dcf_maxpool3x3s2_fwd_idx(int dtype, const void *x, void *y, uint32_t *idx, int B, int H, int W, int Ho, int Wo, int C,
                                        dcf_stream_t stream)
{
    DCF_REQUIRE(x && y && idx && C % 4 == 0 && Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "dcf_maxpool3x3s2_fwd_idx: bad arguments");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("maxpool_fwd", ((double)B * H * W * C + (double)total * 4) * sizeof(T), s, hipLaunchKernelGGL(k_maxpool_fwd_idx<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const T *)x, (T *)y, idx, B, H, W, Ho, Wo, C / 4)); })
    return DCF_OK;
}

extern "C" int dcf_maxpool3x3s2_bwd_idx(int dtype, const uint32_t *idx, const void *gy, void *gx, int B, int H, int W, int Ho, int Wo, int C,
                                        dcf_stream_t stream)
{
    DCF_REQUIRE(idx && gy && gx && C % 4 == 0, "dcf_maxpool3x3s2_bwd_idx: bad arguments");
    const int64_t total = (int64_t)B * H * W * (C / 4);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("maxpool_bwd", ((double)B * Ho * Wo * C + (double)total * 4) * sizeof(T), s, hipLaunchKernelGGL(k_maxpool_bwd_idx<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, idx, (const T *)gy, (T *)gx, B, H, W, Ho, Wo, C / 4)); })
    return DCF_OK;
}

extern "C" int dcf_maxpool3x3s2_bwd(int dtype, const void *x, const void *y, const void *gy, void *gx, int B, int H, int W,
                                    int Ho, int Wo, int C, dcf_stream_t stream)
{
    DCF_REQUIRE(x && gy && gx && C % 4 == 0, "dcf_maxpool3x3s2_bwd: bad arguments");
    const int64_t total = (int64_t)B * H * W * (C / 4);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, {
        if (y) DCF_LAUNCH_B("maxpool_bwd", ((double)B * Ho * Wo * C + (double)total * 4) * sizeof(T), s, hipLaunchKernelGGL(k_maxpool_bwd_y<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const T *)x, (const T *)y, (const T *)gy, (T *)gx, B, H, W, Ho, Wo, C / 4));
        else DCF_LAUNCH_B("maxpool_bwd", ((double)B * Ho * Wo * C + (double)total * 4) * sizeof(T), s, hipLaunchKernelGGL(k_maxpool_bwd<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const T *)x, (const T *)gy, (T *)gx, B, H, W, Ho, Wo, C / 4));
    })
    return DCF_OK;
}

extern "C" int dcf_head_fwd(int dtype, const void *head, int Cp, const float *anchors, float *pred, int B, int h, int w,
                            dcf_stream_t stream)
{
    DCF_REQUIRE(head && anchors && pred && Cp >= 20 && Cp % 4 == 0, "dcf_head_fwd: bad arguments (Cp=%d)", Cp);
    const int hw = h * w;
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("head_fwd", (double)B * hw * (Cp * sizeof(T) + 32 * 4.0), s, hipLaunchKernelGGL(k_head_fwd<T>, dim3(cdiv((int64_t)B * hw, 256)), dim3(256), 0, s, (const T *)head, Cp, anchors, pred, B, hw)); })
    return DCF_OK;
}

extern "C" int dcf_head_bwd(int dtype, const void *head, int Cp, const float *anchors, const float *pred, const float *gpred,
                            void *ghead, int B, int h, int w, dcf_stream_t stream)
{
    (void)head;
    DCF_REQUIRE(anchors && pred && gpred && ghead && Cp >= 20 && Cp % 4 == 0, "dcf_head_bwd: bad arguments");
    const int hw = h * w;
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("head_bwd", (double)B * hw * (Cp * sizeof(T) + 2 * 32 * 4.0), s, hipLaunchKernelGGL(k_head_bwd<T>, dim3(cdiv((int64_t)B * hw, 256)), dim3(256), 0, s, Cp, anchors, pred, gpred, (T *)ghead, B, hw)); })
    return DCF_OK;
}

extern "C" int dcf_weight_prep(int dtype, const dcf_conv_param *table, int nconv, const float *params, const float *buffers,
                               void *warena, float *ssarena, float eps, dcf_stream_t stream)
{
    DCF_REQUIRE(table && nconv > 0 && nconv <= DCF_PREP_MAXCONV && params && warena && ssarena, "dcf_weight_prep: bad arguments (1..%d layers)", DCF_PREP_MAXCONV);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH("weight_prep", s, hipLaunchKernelGGL(k_weight_prep<T>, dim3(2048), dim3(256), 0, s, table, nconv, params, buffers, (char *)warena, ssarena, eps)); })
    return DCF_OK;
}

static int wgrad_finalize_impl(const char *who, const dcf_conv_param *table, int nconv, int max_cout, const int32_t *cout_host, const float *params,
                               const float *buffers, const float *ssarena, const float *slabs, const float *gsum, float *grads, float eps,
                               dcf_stream_t stream)
{
    DCF_REQUIRE(table && nconv > 0 && params && ssarena && slabs && gsum && grads, "%s: bad arguments", who);
    hipStream_t s = S(stream);
    static DcfOpt grp_env_o("FINALIZE_GROUPS"); const char *grp_env = grp_env_o.str();
    const int grouped = grp_env ? atoi(grp_env) : 1;
    FinRows rows;
    rows.n = 0;
    dim3 grid(max_cout, nconv);
    if (cout_host && nconv <= DCF_FIN_MAXCONV) {
        int tot = 0;
        for (int i = 0; i < nconv; ++i) { rows.first[i] = tot; tot += cout_host[i]; }
        for (int i = nconv; i <= DCF_FIN_MAXCONV; ++i) rows.first[i] = tot;
        rows.n = nconv;
        grid = dim3(tot);
    }
    DCF_REQUIRE(grid.x > 0, "%s: empty grid", who);
    DCF_LAUNCH("wgrad_finalize", s, hipLaunchKernelGGL(k_wgrad_finalize, grid, dim3(256), 0, s, table, params, buffers, ssarena, slabs, gsum, grads, eps, grouped, rows));
    return DCF_OK;
}

extern "C" int dcf_wgrad_finalize(const dcf_conv_param *table, int nconv, int max_cout, const float *params, const float *buffers,
                                  const float *ssarena, const float *slabs, const float *gsum, float *grads, float eps, dcf_stream_t stream)
{
    DCF_REQUIRE(max_cout > 0, "dcf_wgrad_finalize: bad arguments");
    return wgrad_finalize_impl("dcf_wgrad_finalize", table, nconv, max_cout, nullptr, params, buffers, ssarena, slabs, gsum, grads, eps, stream);
}

extern "C" int dcf_wgrad_finalize_rows(const dcf_conv_param *table, int nconv, const int32_t *cout_host, const float *params, const float *buffers,
                                       const float *ssarena, const float *slabs, const float *gsum, float *grads, float eps, dcf_stream_t stream)
{
    DCF_REQUIRE(cout_host, "dcf_wgrad_finalize_rows: bad arguments");
    int mx = 1;
    for (int i = 0; i < nconv; ++i) mx = cout_host[i] > mx ? cout_host[i] : mx;
    return wgrad_finalize_impl("dcf_wgrad_finalize_rows", table, nconv, mx, cout_host, params, buffers, ssarena, slabs, gsum, grads, eps, stream);
}

extern "C" int dcf_adam_step(float *params, const float *grads, float *m, float *v, int64_t n, float lr, float beta1,
                             float beta2, float eps, int step, float gscale, dcf_stream_t stream)
{
    DCF_REQUIRE(params && grads && m && v && n >= 0 && step >= 1, "dcf_adam_step: bad arguments");
    if (n == 0) return DCF_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipStream_t s = S(stream);
    DCF_LAUNCH_B("adam", (double)n * 28.0, s, hipLaunchKernelGGL(k_adam, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, s, params, grads, m, v, n, (float)(lr / bc1), beta1, beta2,
                                             (float)(1.0 / sqrt(bc2)), eps, gscale));
    return DCF_OK;
}

extern "C" int dcf_cast(int dtype_src, const void *src, int dtype_dst, void *dst, int64_t n, dcf_stream_t stream)
{
    DCF_REQUIRE(src && dst && n % 4 == 0, "dcf_cast: n must be a multiple of 4");
    hipStream_t s = S(stream);
    const int64_t n4 = n / 4;
    if (n4 == 0) return DCF_OK;
    dim3 g(cdiv(n4, 256)), b(256);
#define DCF_CAST(TS_, TD_) DCF_LAUNCH_B("cast", (double)n4 * 4 * (sizeof(TS_) + sizeof(TD_)), s, hipLaunchKernelGGL((k_cast<TS_, TD_>), g, b, 0, s, (const TS_ *)src, (TD_ *)dst, n4))
    if (dtype_src == DCF_F32 && dtype_dst == DCF_BF16) DCF_CAST(float, bf16_t);
    else if (dtype_src == DCF_BF16 && dtype_dst == DCF_F32) DCF_CAST(bf16_t, float);
    else if (dtype_src == DCF_F32 && dtype_dst == DCF_F32) DCF_CAST(float, float);
    else if (dtype_src == DCF_BF16 && dtype_dst == DCF_BF16) DCF_CAST(bf16_t, bf16_t);
    else if (dtype_src == DCF_F32 && dtype_dst == DCF_F16) DCF_CAST(float, f16_t);
    else if (dtype_src == DCF_F16 && dtype_dst == DCF_F32) DCF_CAST(f16_t, float);
    else if (dtype_src == DCF_F16 && dtype_dst == DCF_F16) DCF_CAST(f16_t, f16_t);
    else { dcf_set_error("dcf_cast: unsupported dtype pair"); return DCF_EUNSUPPORTED; }
#undef DCF_CAST
    return DCF_OK;
}

extern "C" int dcf_rowscale_bias_fwd(int dtype, void *y, const float *cnt, const float *b2, int64_t npix, int C, dcf_stream_t stream)
{
    DCF_REQUIRE(y && cnt && b2 && C % 4 == 0, "dcf_rowscale_bias_fwd: bad arguments");
    const int64_t nvec = npix * (C / 4);
    if (nvec == 0) return DCF_OK;
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("rowscale_bias_fwd", (double)nvec * 8 * sizeof(T) + npix * 4.0, s, hipLaunchKernelGGL(k_rowscale_bias_fwd<T>, dim3(cdiv(nvec, 256)), dim3(256), 0, s, (T *)y, cnt, b2, nvec, C / 4)); })
    return DCF_OK;
}

extern "C" int dcf_rowscale_bias_bwd(int dtype, const void *gy, const float *cnt, float *gb2, int64_t npix, int C, dcf_stream_t stream)
{
    DCF_REQUIRE(gy && cnt && gb2 && C % 4 == 0, "dcf_rowscale_bias_bwd: bad arguments");
    const int V = C % 8 == 0 ? 8 : 4;
    const int cg = C / V;
    const int64_t nvec = npix * cg;
    if (nvec == 0) return DCF_OK;
    int blocks;
    // every workgroup ends with C same-address atomics on gb2: 256 workgroups instead of 1024 (0.118 -> 0.068 ms per step)
    const int64_t stride = chan_stride(nvec, cg, blocks, 64);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, {
        const double bytes = (double)npix * C * sizeof(T) + npix * 4.0;
        if (V == 8) DCF_LAUNCH_B("rowscale_bias_bwd", bytes, s, hipLaunchKernelGGL((k_rowscale_bias_bwd<T, 8>), dim3(blocks), dim3(256), sizeof(float) * C, s, (const T *)gy, cnt, gb2, nvec, cg, stride));
        else DCF_LAUNCH_B("rowscale_bias_bwd", bytes, s, hipLaunchKernelGGL((k_rowscale_bias_bwd<T, 4>), dim3(blocks), dim3(256), sizeof(float) * C, s, (const T *)gy, cnt, gb2, nvec, cg, stride));
    })
    return DCF_OK;
}

extern "C" int dcf_relu_mask_rowscale_bwd(int dtype, const void *gy, const void *y, const float *cnt, void *gout, float *gb2, int64_t npix,
                                          int C, dcf_stream_t stream)
{
    DCF_REQUIRE(gy && y && cnt && gout && gb2 && gout != gy && C % 4 == 0, "dcf_relu_mask_rowscale_bwd: bad arguments (gout must be its own tensor)");
    const int V = C % 8 == 0 ? 8 : 4;
    const int cg = C / V;
    const int64_t nvec = npix * cg;
    if (nvec == 0) return DCF_OK;
    int blocks;
    const int64_t stride = chan_stride(nvec, cg, blocks, 64);
    hipStream_t s = S(stream);
    DCF_DISPATCH_DTYPE(dtype, {
        const double bytes = (double)npix * C * sizeof(T) * 3 + npix * 4.0;
        if (V == 8) DCF_LAUNCH_B("relu_mask_rowscale_bwd", bytes, s, hipLaunchKernelGGL((k_relu_mask_rowscale_bwd<T, 8>), dim3(blocks), dim3(256), sizeof(float) * 256 * 8, s, (const T *)gy, (const T *)y, cnt, (T *)gout, gb2, nvec, cg, stride));
        else DCF_LAUNCH_B("relu_mask_rowscale_bwd", bytes, s, hipLaunchKernelGGL((k_relu_mask_rowscale_bwd<T, 4>), dim3(blocks), dim3(256), sizeof(float) * 256 * 4, s, (const T *)gy, (const T *)y, cnt, (T *)gout, gb2, nvec, cg, stride));
    })
    return DCF_OK;
}

// ------------------------------------------------------------------ train-mode BatchNorm C ABI
static inline int bn_blocks(int64_t nvec, int cg, int64_t *stride)
{
    // <= 512 workgroups of partials; four vectors per thread where the tensor is small (a quarter of the partial rows for the
    // finalisation kernel to walk)
    int64_t want = std::min<int64_t>((nvec + 3) / 4, 128 * 1024);
    if (want < cg) want = cg;
    *stride = want / cg * cg;
    return cdiv(*stride, 256);
}

extern "C" size_t dcf_bn_workspace_bytes(int C) { return sizeof(float) * 2 * (size_t)C * 512; }

extern "C" int dcf_bn_train_fwd(int dtype, const void *x, const float *gamma, const float *beta, const void *res, void *y,
                                float *mean, float *invstd, float *running_mean, float *running_var, int64_t npix, int C,
                                float eps, float momentum, int relu, void *ws, dcf_stream_t stream)
{
    DCF_REQUIRE(x && gamma && beta && y && mean && invstd && ws && C % 4 == 0 && npix > 0, "dcf_bn_train_fwd: bad arguments");
    const int V = C % 8 == 0 ? 8 : 4;
    const int cg = C / V;
    const int64_t nvec = npix * cg;
    int64_t stride;
    const int nblk = bn_blocks(nvec, cg, &stride);
    hipStream_t s = S(stream);
    float *partial = (float *)ws;
#define DCF_BN_FWD(V_)                                                                                                                             \
    do {                                                                                                                                           \
        DCF_LAUNCH("bn_stats_partial", s, hipLaunchKernelGGL((k_bn_partial<T, false, V_>), dim3(nblk), dim3(256), 0, s, (const T *)x, \
                                                              (const T *)nullptr, (const float *)nullptr, (const float *)nullptr, partial, nvec, cg, stride)); \
        DCF_LAUNCH("bn_stats_final", s, hipLaunchKernelGGL(k_bn_stats_final, dim3(C), dim3(64), 0, s, partial, nblk, C, (double)npix, eps,         \
                                                            momentum, mean, invstd, running_mean, running_var));                                   \
        DCF_LAUNCH("bn_apply_fwd", s, hipLaunchKernelGGL((k_bn_apply_fwd<T, V_>), dim3(cdiv(nvec, 256)), dim3(256), 0, s, (const T *)x, mean, invstd, gamma, beta, \
                                                          (const T *)res, (T *)y, nvec, cg, relu));                                                \
    } while (0)
    DCF_DISPATCH_DTYPE(dtype, {
        if (V == 8) DCF_BN_FWD(8); else DCF_BN_FWD(4);
    })
#undef DCF_BN_FWD
    return DCF_OK;
}

extern "C" int dcf_bn_train_bwd(int dtype, const void *g, const void *x, const float *mean, const float *invstd, const float *gamma,
                                float *dgamma, float *dbeta, void *dx, int64_t npix, int C, void *ws, dcf_stream_t stream)
{
    DCF_REQUIRE(g && x && mean && invstd && gamma && dgamma && dbeta && dx && ws && C % 4 == 0 && npix > 0, "dcf_bn_train_bwd: bad arguments");
    const int V = C % 8 == 0 ? 8 : 4;
    const int cg = C / V;
    const int64_t nvec = npix * cg;
    int64_t stride;
    const int nblk = bn_blocks(nvec, cg, &stride);
    hipStream_t s = S(stream);
    float *partial = (float *)ws;
#define DCF_BN_BWD(V_)                                                                                                                             \
    do {                                                                                                                                           \
        DCF_LAUNCH("bn_bwd_partial", s, hipLaunchKernelGGL((k_bn_partial<T, true, V_>), dim3(nblk), dim3(256), 0, s, (const T *)x, (const T *)g, \
                                                            mean, invstd, partial, nvec, cg, stride));                                             \
        DCF_LAUNCH("bn_bwd_final", s, hipLaunchKernelGGL(k_bn_bwd_final, dim3(C), dim3(64), 0, s, partial, nblk, C, dgamma, dbeta));               \
        DCF_LAUNCH("bn_apply_bwd", s, hipLaunchKernelGGL((k_bn_apply_bwd<T, V_>), dim3(cdiv(nvec, 256)), dim3(256), 0, s, (const T *)g, (const T *)x, mean, invstd, gamma, \
                                                          dgamma, dbeta, (T *)dx, nvec, cg, 1.0f / (float)npix));                                  \
    } while (0)
    DCF_DISPATCH_DTYPE(dtype, {
        if (V == 8) DCF_BN_BWD(8); else DCF_BN_BWD(4);
    })
#undef DCF_BN_BWD
    return DCF_OK;
}
