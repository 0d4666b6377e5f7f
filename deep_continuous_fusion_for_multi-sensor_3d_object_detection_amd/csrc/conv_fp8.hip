// conv_fp8.hip -- forward convolutions with fp8 (OCP e4m3) operands on v_mfma_scale_f32_32x32x64_f8f6f4
// (BASELINE.json configs[4], SURVEY.md 8(d) cfg5: "fp8 MFMA convs, fp32 accumulate, bf16 epilogue").
//
// Numerics: y[m][co] = act( (sum_k q(x[m][k] * sx) * q(w[co][k] * sw[co])) / (sx * sw[co]) + shift[co] + res[m][co] )
//   q()  = round-to-nearest-even to e4m3 (v_cvt_pk_fp8_f32), products exact in fp32, fp32 accumulation
//   sw   = 448 / max_k |w[co][k]| per output channel (dcf_weight_prep_fp8), sx = a power of two derived from the
//          tensor's absolute maximum one step earlier (delayed scaling; f8_act_scale) -- both live in HBM, no host sync.
// The block-scale operands of the MFMA are the constant 2^0: the scaling above is per tensor / per channel, applied in
// the epilogue.  Storage of activations stays bf16/fp16 (residuals, the backward); the fp8 image of a conv's input is
// written by dcf_cast_fp8 right before the conv.  Only the forward uses fp8: the backward convolutions read the saved
// 16-bit activations (straight-through).
#include "dcf_common.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// power-of-two scale that maps an absolute maximum of `amax` to at most 224 (half the e4m3 range of 448: the amax is
// the previous step's).  Exact bit arithmetic, so every kernel (and the CPU oracle) derives the same value.
__host__ __device__ __forceinline__ float f8_act_scale(float amax)
{
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    const float s = 224.f / amax;
    unsigned u;
    memcpy(&u, &s, 4);
    u &= 0x7F800000u;
    if (u == 0u) return 1.1754944e-38f;          // 224/amax below the normal range: smallest normal power of two
    float r;
    memcpy(&r, &u, 4);
    return r;
}

// value of v after a round trip through the storage type
__device__ __forceinline__ float stored(float v, const float *) { return v; }
__device__ __forceinline__ float stored(float v, const bf16_t *) { return bf2f(f2bf(v)); }
__device__ __forceinline__ float stored(float v, const f16_t *) { return h2f(f2h(v)); }

__device__ __forceinline__ float clamp448(float v) { return fminf(fmaxf(v, -448.f), 448.f); }
__device__ __forceinline__ unsigned pack4_f8(float a, float b, float c, float d)
{
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(a), clamp448(b), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(c), clamp448(d), w, true);
    return (unsigned)w;
}

// ------------------------------------------------------------------------------------ activation cast
// x8 = q(x * sx), sx = f8_act_scale(*amax_prev); amax_cur[64] = partial maxima of |x| (slot = workgroup & 63: one hot
// address would serialise the workgroups' atomics).  8 elements per thread per pass.
template <typename T>
__global__ void __launch_bounds__(256) k_cast_f8(const T *x, unsigned *x8, const float *amax_prev, unsigned *amax_cur, int64_t n8)
{
    __shared__ float wmax[4];
    const float sx = f8_act_scale(amax_prev ? *amax_prev : 0.f);
    float am = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = ld4(x + i * 8), b = ld4(x + i * 8 + 4);
        am = fmaxf(am, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        am = fmaxf(am, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
        uint2 o;
        o.x = pack4_f8(a.x * sx, a.y * sx, a.z * sx, a.w * sx);
        o.y = pack4_f8(b.x * sx, b.y * sx, b.z * sx, b.w * sx);
        *reinterpret_cast<uint2 *>(x8 + i * 2) = o;
    }
    if (!amax_cur) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        am = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (am > 0.f && am < 3.0e38f) atomicMax(amax_cur + (blockIdx.x & 63), __float_as_uint(am));     // non-negative floats order like their bits
    }
}

// ------------------------------------------------------------------------------------ weight images
// One workgroup per (output channel, conv): w8[co][k] = q(bn_scale*W[co][k] * 448/amax_co), wscale[co] = amax_co/448.
// Workgroup (0, conv) also rolls the conv's activation maxima (DCF_F8_AMAX_STRIDE floats per conv: 64 partial maxima
// of the current step, then the previous step's maximum): prev <- max(cur[0..63]), cur <- 0.
__global__ void __launch_bounds__(256) k_weight_prep_f8(const dcf_conv_param *table, const dcf_f8_param *f8tab, const float *params,
                                                        const float *buffers, char *w8arena, float *wsarena, float *amax, float eps)
{
    __shared__ float wmax[4];
    const dcf_conv_param d = table[blockIdx.y];
    const dcf_f8_param f = f8tab[blockIdx.y];
    if (f.w8_off < 0) return;
    const int co = blockIdx.x;
    if (co == 0 && threadIdx.x < 64) {
        float *slot = amax + (int64_t)DCF_F8_AMAX_STRIDE * blockIdx.y;
        float m = slot[threadIdx.x];
        slot[threadIdx.x] = 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (threadIdx.x == 0) slot[64] = m;
    }
    if (co >= d.cout_pad) return;
    const int K = d.taps * d.cin;
    unsigned *dst = reinterpret_cast<unsigned *>(w8arena + f.w8_off + (int64_t)co * K);
    if (co >= d.cout) {                                   // padded output channels: zero weights
        for (int k = threadIdx.x; k < K / 4; k += 256) dst[k] = 0u;
        if (threadIdx.x == 0) wsarena[f.wscale_off + co] = 0.f;
        return;
    }
    float bn = 1.f;
    if (d.gamma_off >= 0) bn = params[d.gamma_off + co] * rsqrtf(buffers[d.var_off + co] + eps);
    const float *src = params + d.w_off + (int64_t)co * K;
    float am = 0.f;
    for (int k = threadIdx.x; k < K / 4; k += 256) {
        const float4 v = ld4(src + k * 4);
        am = fmaxf(am, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])) * fabsf(bn);
    const float sw = am > 0.f ? 448.f / am : 1.f;
    const float m = bn * sw;
    for (int k = threadIdx.x; k < K / 4; k += 256) {
        const float4 v = ld4(src + k * 4);
        dst[k] = pack4_f8(v.x * m, v.y * m, v.z * m, v.w * m);
    }
    if (threadIdx.x == 0) wsarena[f.wscale_off + co] = am > 0.f ? am / 448.f : 1.f;
}

// ------------------------------------------------------------------------------------ the convolution
struct Conv8Args {
    const char *x;        // fp8 [B][Hi][Wi][Ck]
    const char *w;        // fp8 [Cn][taps][Ck]
    const float *wscale;  // [Cn]
    const float *xamax;   // device scalar the activation scale derives from (null: scale 1)
    const float *shift;   // [Cn] or null
    const char *res;      // [M][Cn] (TO) or null
    char *y;              // [M][Cn] (TO)
    int B, Hi, Wi, Ck, Ho, Wo, Cn, kh, kw, stride, pad, relu, M;
    unsigned xbytes, wbytes;
    // optional second output: the fp8 image of y for the convolution that consumes it (saves that conv's cast pass)
    char *y8;                 // [M][Cn] e4m3 or null
    const float *y8amax;      // device scalar the image's scale derives from (null: 1)
    unsigned *y8cur;          // 64 partial maxima of |y| (this step), or null
};

// Block = 256 threads = WN x WM waves; wave tile TN*32 channels x TM*32 pixels; K walked tap by tap in chunks of KB
// channels (= bytes).  Two LDS stages, one barrier per chunk: chunk it+1 is written to the other stage and chunk it+2
// requested from L2 while the MFMAs of chunk it run.  LDS rows are KB+16 bytes: the ds_read_b128 pairs of the 32-byte
// fragments are conflict free.  Both operands use the same lane -> k map (32 consecutive bytes per lane half), so the
// order of k inside one K=64 instruction does not matter.
template <typename TO, int KB, int TN, int TM, int WN, int WM>
__global__ void __launch_bounds__(256) k_conv_f8(Conv8Args a)
{
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    constexpr int PITCH = KB + 16;
    constexpr int CPR = KB / 16;
    constexpr int NCW = BN * CPR, NCX = BM * CPR;
    constexpr int NLW = (NCW + 255) / 256, NLX = (NCX + 255) / 256;
    constexpr int STAGE = (BN + BM) * PITCH;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid / WM, wm = wid % WM;
    const int r = lane & 31, h = lane >> 5;
    // XCD-aware tile order: XCD x takes the x-th contiguous chunk of the (pixel tile, channel tile) list
    const int nt = a.Cn / BN;
    const int mtiles = (a.M + BM - 1) / BM;
    const int nblk = mtiles * nt;
    const int chunk = (nblk + 7) >> 3;
    const int gidx = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (gidx >= nblk) return;
    const int n0 = (gidx % nt) * BN;
    const int m0 = (gidx / nt) * BM;
    const int taps = a.kh * a.kw;
    const int cchunks = a.Ck / KB;

    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcW = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;          // past the end of either buffer: reads as zero

    int xb[NLX], xh[NLX], xw[NLX];
    unsigned xoff[NLX], woff[NLW];
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
        const int c = tid + i * 256;
        const int m = m0 + c / CPR;
        xoff[i] = (c % CPR) * 16;
        if ((NCX % 256 == 0 || c < NCX) && m < a.M) {
            const int b = m / (a.Ho * a.Wo);
            const int rem = m - b * (a.Ho * a.Wo);
            const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            xb[i] = b * a.Hi * a.Wi;
            xh[i] = oh * a.stride - a.pad;
            xw[i] = ow * a.stride - a.pad;
        } else {
            xb[i] = -1; xh[i] = 0; xw[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int c = tid + i * 256;
        woff[i] = (NCW % 256 == 0 || c < NCW) ? (unsigned)(n0 + c / CPR) * (unsigned)(taps * a.Ck) + (c % CPR) * 16 : OOB;
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    unsigned pix[NLX];
    auto set_tap = [&](int ki, int kj) {
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            const int ih = xh[i] + ki, iw = xw[i] + kj;
            const bool ok = xb[i] >= 0 && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
            pix[i] = ok ? (unsigned)(xb[i] + ih * a.Wi + iw) * (unsigned)a.Ck + xoff[i] : OOB;
        }
    };
    uint4 rw[NLW], rx[NLX];
    auto load_global = [&](unsigned koff, unsigned ccoff) {
#pragma unroll
        for (int i = 0; i < NLW; ++i)
            rw[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srcW, woff[i] == OOB ? OOB : woff[i] + koff, 0, 0));
#pragma unroll
        for (int i = 0; i < NLX; ++i)
            rx[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srcX, pix[i] == OOB ? OOB : pix[i] + ccoff, 0, 0));
    };
    auto store_lds = [&](int buf) {
        char *ldsW = lds + buf * STAGE, *ldsX = ldsW + BN * PITCH;
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const int c = tid + i * 256;
            if (NCW % 256 == 0 || c < NCW) *reinterpret_cast<uint4 *>(ldsW + (c / CPR) * PITCH + (c % CPR) * 16) = rw[i];
        }
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            const int c = tid + i * 256;
            if (NCX % 256 == 0 || c < NCX) *reinterpret_cast<uint4 *>(ldsX + (c / CPR) * PITCH + (c % CPR) * 16) = rx[i];
        }
    };
    const int nit = taps * cchunks;
    int ki = 0, kj = 0, cc = 0;
    auto advance = [&]() {
        if (++cc == cchunks) {
            cc = 0;
            if (++kj >= a.kw) { kj = 0; ++ki; }
            set_tap(ki, kj);
        }
    };
    auto koff = [&]() { return (unsigned)((ki * a.kw + kj) * cchunks + cc) * KB; };
    constexpr int ONE = 0x7F7F7F7F;                  // E8M0 block scale 2^0 for every K block of both operands
    auto compute = [&](int buf) {
        const char *ldsW = lds + buf * STAGE, *ldsX = ldsW + BN * PITCH;
#pragma unroll
        for (int ks = 0; ks < KB / 64; ++ks) {
            v8i fa[TN], fb[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const char *p = ldsW + ((wn * TN + i) * 32 + r) * PITCH + ks * 64 + h * 32;
                const uint4 lo = *reinterpret_cast<const uint4 *>(p), hi = *reinterpret_cast<const uint4 *>(p + 16);
                fa[i] = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const char *p = ldsX + ((wm * TM + j) * 32 + r) * PITCH + ks * 64 + h * 32;
                const uint4 lo = *reinterpret_cast<const uint4 *>(p), hi = *reinterpret_cast<const uint4 *>(p + 16);
                fb[j] = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[i], fb[j], acc[i][j], 0, 0, 0, ONE, 0, ONE);
        }
    };
    if (nit > 0) {
        set_tap(0, 0);
        load_global(0, 0);
        store_lds(0);
    }
    if (nit > 1) { advance(); load_global(koff(), (unsigned)cc * KB); }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        compute(it & 1);
        if (it + 1 < nit) {
            store_lds((it + 1) & 1);
            if (it + 2 < nit) { advance(); load_global(koff(), (unsigned)cc * KB); }
        }
        __syncthreads();
    }

    // epilogue: the MFMA leaves lane (pixel r, half h) with channels 8q+4h+{0..3} of each 32-channel tile; after acc_rows8
    // it holds channels 16p+8h+{0..7} in registers 8p..8p+7 (16-byte stores, 8-byte fp8 stores)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc_rows8(acc[i][j]);
    const float inv_sx = 1.f / f8_act_scale(a.xamax ? *a.xamax : 0.f);
    const float sy = a.y8 ? f8_act_scale(a.y8amax ? *a.y8amax : 0.f) : 0.f;
    float am = 0.f;
    TO *y = reinterpret_cast<TO *>(a.y);
    const TO *res = reinterpret_cast<const TO *>(a.res);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + (wm * TM + j) * 32 + r;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int c = n0 + (wn * TN + i) * 32 + 16 * p + 8 * h;
                float v[8], ws[8];
                ld8(a.wscale + c, ws);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = acc[i][j][8 * p + k] * (ws[k] * inv_sx);
                if (a.shift) {
                    float s[8];
                    ld8(a.shift + c, s);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += s[k];
                }
                const size_t o = (size_t)m * a.Cn + c;
                if (res) {
                    float rr[8];
                    ld8(res + o, rr);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += rr[k];
                }
                if (a.relu) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                }
                st8(y + o, v);
                if (a.y8) {
                    // the image is made from the value as stored (rounded to TO): identical to dcf_cast_fp8 of y
                    float t[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { t[k] = stored(v[k], y); am = fmaxf(am, fabsf(t[k])); }
                    uint2 q8;
                    q8.x = pack4_f8(t[0] * sy, t[1] * sy, t[2] * sy, t[3] * sy);
                    q8.y = pack4_f8(t[4] * sy, t[5] * sy, t[6] * sy, t[7] * sy);
                    *reinterpret_cast<uint2 *>(a.y8 + o) = q8;
                }
            }
        }
    }
    if (a.y8 && a.y8cur) {
        __shared__ float wmax[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
        if (lane == 0) wmax[wid] = am;
        __syncthreads();
        if (tid == 0) {
            am = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (am > 0.f && am < 3.0e38f) atomicMax(a.y8cur + (blockIdx.x & 63), __float_as_uint(am));
        }
    }
}

template <typename TO>
int launch_f8(const Conv8Args &a, hipStream_t s, double flops)
{
    char name[64];
#define DCF_F8(KB_, TN_, TM_, WN_, WM_)                                                                             \
    do {                                                                                                            \
        constexpr int BN_ = WN_ * TN_ * 32, BM_ = WM_ * TM_ * 32;                                                   \
        dim3 grid((((int64_t)cdiv(a.M, BM_) * (a.Cn / BN_) + 7) / 8) * 8);                                          \
        snprintf(name, sizeof(name), "conv_fwd_fp8<%d,%d,%d,%d,%d>", KB_, TN_, TM_, WN_, WM_);                       \
        DCF_LAUNCH_W(name, flops, s, hipLaunchKernelGGL((k_conv_f8<TO, KB_, TN_, TM_, WN_, WM_>), grid, dim3(256), 0, s, a)); \
        return DCF_OK;                                                                                              \
    } while (0)
    // KB = 64 everywhere: two stages of a 128x128 (or 64x256) tile stay under the 64-KB static LDS limit
    static DcfOpt force_env_o("F8_TILE"); const char *force_env = force_env_o.str();
    const int force = force_env ? atoi(force_env) : -1;
    auto blocks = [&](int bn, int bm) { return (int64_t)cdiv(a.M, bm) * (a.Cn / bn); };
    // the biggest tile that still gives the chip >= ~2 workgroups per CU
    if ((force == 0 || (force < 0 && blocks(128, 128) >= 512)) && a.Cn % 128 == 0) {
        DCF_F8(64, 2, 2, 2, 2);
    } else if ((force == 1 || (force < 0 && blocks(64, 256) >= 512)) && a.Cn % 64 == 0) {
        DCF_F8(64, 2, 2, 1, 4);
    } else if ((force == 2 || (force < 0 && blocks(64, 128) >= 256)) && a.Cn % 64 == 0) {
        DCF_F8(64, 2, 1, 1, 4);
    } else if (a.Cn % 64 == 0) {
        DCF_F8(64, 1, 1, 2, 2);
    } else {
        DCF_F8(64, 1, 1, 1, 4);
    }
#undef DCF_F8
}

}  // namespace

// ================================================================== C ABI
extern "C" int dcf_fp8_act_scale(float amax, float *scale)
{
    DCF_REQUIRE(scale, "dcf_fp8_act_scale: null pointer");
    *scale = f8_act_scale(amax);
    return DCF_OK;
}

extern "C" int dcf_cast_fp8(int dtype, const void *x, void *x8, const float *amax_prev, float *amax_cur, int64_t n, dcf_stream_t stream)
{
    DCF_REQUIRE(x && x8 && n % 8 == 0, "dcf_cast_fp8: n must be a multiple of 8");
    DCF_REQUIRE(dtype == DCF_BF16 || dtype == DCF_F16 || dtype == DCF_F32, "dcf_cast_fp8: unsupported dtype %d", dtype);
    if (n == 0) return DCF_OK;
    hipStream_t s = S(stream);
    const int64_t n8 = n / 8;
    const int blocks = (int)std::min<int64_t>(cdiv(n8, 256), 2048);      // one atomic per workgroup, spread over 64 addresses
    DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH("cast_fp8", s, hipLaunchKernelGGL(k_cast_f8<T>, dim3(blocks), dim3(256), 0, s, (const T *)x, (unsigned *)x8, amax_prev, (unsigned *)amax_cur, n8)); })
    return DCF_OK;
}

extern "C" int dcf_weight_prep_fp8(const dcf_conv_param *table, const dcf_f8_param *f8table, int nconv, int max_cout_pad, const float *params,
                                   const float *buffers, void *w8arena, float *wsarena, float *amax, float eps, dcf_stream_t stream)
{
    DCF_REQUIRE(table && f8table && nconv > 0 && max_cout_pad > 0 && params && w8arena && wsarena && amax, "dcf_weight_prep_fp8: bad arguments");
    hipStream_t s = S(stream);
    DCF_LAUNCH("weight_prep_fp8", s, hipLaunchKernelGGL(k_weight_prep_f8, dim3(max_cout_pad, nconv), dim3(256), 0, s, table, f8table, params, buffers,
                                                        (char *)w8arena, wsarena, amax, eps));
    return DCF_OK;
}

extern "C" int dcf_conv2d_fwd_fp8(int out_dtype, const void *x8, const void *w8, const float *wscale, const float *xamax, const float *shift,
                                  const void *res, void *y, void *y8, const float *y8amax, float *y8cur, int B, int H, int W, int Cin,
                                  int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int relu, dcf_stream_t stream)
{
    DCF_REQUIRE(out_dtype == DCF_BF16 || out_dtype == DCF_F16 || out_dtype == DCF_F32, "dcf_conv2d_fwd_fp8: unsupported output dtype %d", out_dtype);
    DCF_REQUIRE(x8 && w8 && wscale && y, "dcf_conv2d_fwd_fp8: null pointer");
    DCF_REQUIRE(Cin % 64 == 0, "dcf_conv2d_fwd_fp8: Cin must be a multiple of 64 (one K=64 MFMA step), got %d", Cin);
    DCF_REQUIRE(Cout % 32 == 0, "dcf_conv2d_fwd_fp8: Cout must be a multiple of 32 (Cout=%d)", Cout);
    DCF_REQUIRE(kh >= 1 && kw >= 1 && kh <= 7 && kw <= 7 && (stride == 1 || stride == 2), "dcf_conv2d_fwd_fp8: unsupported kernel %dx%d stride %d", kh, kw, stride);
    DCF_REQUIRE(Ho == (H + 2 * pad - kh) / stride + 1 && Wo == (W + 2 * pad - kw) / stride + 1, "dcf_conv2d_fwd_fp8: output size mismatch");
    DCF_REQUIRE((int64_t)B * H * W * Cin < 0xFFFFFF00ll, "dcf_conv2d_fwd_fp8: tensor exceeds the 4 GiB buffer-descriptor range");
    Conv8Args a;
    a.x = (const char *)x8; a.w = (const char *)w8; a.wscale = wscale; a.xamax = xamax; a.shift = shift; a.res = (const char *)res; a.y = (char *)y;
    a.y8 = (char *)y8; a.y8amax = y8amax; a.y8cur = (unsigned *)y8cur;
    a.B = B; a.Hi = H; a.Wi = W; a.Ck = Cin; a.Ho = Ho; a.Wo = Wo; a.Cn = Cout;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad; a.relu = relu; a.M = B * Ho * Wo;
    a.xbytes = (unsigned)((int64_t)B * H * W * Cin);
    a.wbytes = (unsigned)((int64_t)Cout * kh * kw * Cin);
    const double flops = 2.0 * a.M * Cout * (double)Cin * kh * kw;
    if (out_dtype == DCF_F32) return launch_f8<float>(a, S(stream), flops);
    if (out_dtype == DCF_F16) return launch_f8<f16_t>(a, S(stream), flops);
    return launch_f8<bf16_t>(a, S(stream), flops);
}
