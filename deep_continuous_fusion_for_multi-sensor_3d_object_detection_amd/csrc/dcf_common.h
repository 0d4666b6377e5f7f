// dcf_common.h -- shared host/device helpers of libdcf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dcf_hip.h"

// ---------------------------------------------------------------- error handling
void dcf_set_error(const char *fmt, ...);

#define DCF_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            dcf_set_error(__VA_ARGS__);        \
            return DCF_EINVAL;                 \
        }                                      \
    } while (0)

#define DCF_HIP(call)                                                                 \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                      \
            dcf_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return DCF_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

// ---------------------------------------------------------------- launch + profiling
// Every kernel launch goes through DCF_LAUNCH so that the optional event timing
// (dcf_prof_enable) brackets it on the stream it is launched on.
void dcf_prof_begin(const char *name, hipStream_t s, double work = 0.0, double bytes = 0.0);
void dcf_prof_end(hipStream_t s);
extern int g_dcf_prof_on;

#define DCF_LAUNCH(name, stream, ...) DCF_LAUNCH_WB(name, 0.0, 0.0, stream, __VA_ARGS__)
// same, with the launch's ALGORITHMIC work recorded next to its time: flops (MFMA-bound kernels) ...
#define DCF_LAUNCH_W(name, work, stream, ...) DCF_LAUNCH_WB(name, work, 0.0, stream, __VA_ARGS__)
// ... and / or bytes: the tensors the kernel has to read and write once, whatever its tiling re-reads (HBM-bound kernels)
#define DCF_LAUNCH_B(name, bytes, stream, ...) DCF_LAUNCH_WB(name, 0.0, bytes, stream, __VA_ARGS__)

// (-DDCF_NO_LAUNCH: a library whose launches do nothing -- tools/host_split.py then times the HOST side of a step alone)
#ifdef DCF_NO_LAUNCH
#define DCF_DO_LAUNCH(...) do { if (g_dcf_prof_on < 0) { __VA_ARGS__; } } while (0)
#else
#define DCF_DO_LAUNCH(...) do { __VA_ARGS__; } while (0)
#endif
#define DCF_LAUNCH_WB(name, work, bytes, stream, ...)                               \
    do {                                                                            \
        if (g_dcf_prof_on) dcf_prof_begin(name, stream, work, bytes);               \
        DCF_DO_LAUNCH(__VA_ARGS__);                                                 \
        if (g_dcf_prof_on) dcf_prof_end(stream);                                    \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            dcf_set_error("launch %s failed: %s", name, hipGetErrorString(e__));    \
            return DCF_ELAUNCH;                                                     \
        }                                                                           \
    } while (0)

// ---------------------------------------------------------------- tuning options
// Every tuning switch of the library is a named option, seeded ONCE per process from the environment variable DCF_<NAME> (on
// first use) and changed afterwards only through dcf_set_option() (tests and tools compare kernels that way).  None of them
// changes results.  Switches that DO -- the timing ablations that turn phases of a kernel off -- exist only in builds with
// -DDCF_ABLATE (make ABLATE=1; tools/rs_ablate.py and friends): the shipped library has no way to reach them.
// Launches come from several host threads (the main thread, the autograd thread, the gradient-bucket hook): a call site's
// cached view is a pair of atomics -- a reader sees either the old value with the old epoch (and refreshes on its next call)
// or the new pair; interned values are never freed, so a stale pointer stays readable.
#include <atomic>
const char *dcf_opt(const char *name);          // current value or nullptr (runtime.cpp; takes the registry mutex)
extern std::atomic<int> g_dcf_opt_epoch;         // bumped by dcf_set_option
struct DcfOpt {                                  // a call site's cached view of one option: static DcfOpt o("RS_KIND");
    const char *name;
    std::atomic<const char *> val;
    std::atomic<int> epoch;
    explicit DcfOpt(const char *n) : name(n), val(nullptr), epoch(-1) {}
    const char *str()
    {
        const int now = g_dcf_opt_epoch.load(std::memory_order_acquire);
        if (epoch.load(std::memory_order_acquire) != now) {
            val.store(dcf_opt(name), std::memory_order_release);
            epoch.store(now, std::memory_order_release);
        }
        return val.load(std::memory_order_acquire);
    }
};
#ifdef DCF_ABLATE
#include <stdlib.h>
static inline int dcf_ablate_opt(const char *name) { const char *e = dcf_opt(name); return e ? atoi(e) : 0; }
#define DCF_DBG(a) ((a).dbg)
#else
static inline int dcf_ablate_opt(const char *) { return 0; }
#define DCF_DBG(a) 0
#endif

static inline hipStream_t S(dcf_stream_t s) { return (hipStream_t)s; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline double esize_of(int dtype) { return dtype == DCF_F32 ? 4.0 : 2.0; }

// ---------------------------------------------------------------- dtype helpers
typedef unsigned short bf16_t;  // raw bits

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f)
{
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

// IEEE half: raw bits in a type of its own (overloads must tell it from bf16_t)
struct f16_t { unsigned short v; };
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float h2f(f16_t h) { return (float)__builtin_bit_cast(_Float16, h.v); }
__device__ __forceinline__ f16_t f2h(float f)
{
    f16_t r;
    r.v = __builtin_bit_cast(unsigned short, (_Float16)f);   // v_cvt_f16_f32: RNE, overflow -> inf
    return r;
}
// two packed 16-bit elements of a 32-bit word -> floats (element 0 in the low half)
template <typename T> __device__ __forceinline__ void unpack2(unsigned w, float &lo, float &hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(unsigned w, float &lo, float &hi)
{
    lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16_t>(unsigned w, float &lo, float &hi)
{
    lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
}

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int size = 4;
    __device__ static __forceinline__ float ld(const float *p) { return *p; }
    __device__ static __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static constexpr int size = 2;
    __device__ static __forceinline__ float ld(const bf16_t *p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t *p, float v) { *p = f2bf(v); }
};

template <> struct DT<f16_t> {
    static constexpr int size = 2;
    __device__ static __forceinline__ float ld(const f16_t *p) { return h2f(*p); }
    __device__ static __forceinline__ void st(f16_t *p, float v) { *p = f2h(v); }
};

// load/store 4 consecutive channels as float4
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t *p)
{
    uint2 u = *reinterpret_cast<const uint2 *>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4(bf16_t *p, float4 v)
{
    uint2 u;
    u.x = pack2bf(v.x, v.y);
    u.y = pack2bf(v.z, v.w);
    *reinterpret_cast<uint2 *>(p) = u;
}

__device__ __forceinline__ float4 ld4(const f16_t *p)
{
    const h16x4 v = *reinterpret_cast<const h16x4 *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(f16_t *p, float4 v)
{
    h16x4 h;
    h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
    *reinterpret_cast<h16x4 *>(p) = h;
}

// load/store 8 consecutive channels (16-byte accesses for the 16-bit types)
__device__ __forceinline__ void ld8(const float *p, float (&v)[8])
{
    const float4 a = ld4(p), b = ld4(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(float *p, const float (&v)[8])
{
    st4(p, make_float4(v[0], v[1], v[2], v[3]));
    st4(p + 4, make_float4(v[4], v[5], v[6], v[7]));
}
__device__ __forceinline__ void ld8(const bf16_t *p, float (&v)[8])
{
    const uint4 u = *reinterpret_cast<const uint4 *>(p);
    unpack2<bf16_t>(u.x, v[0], v[1]); unpack2<bf16_t>(u.y, v[2], v[3]); unpack2<bf16_t>(u.z, v[4], v[5]); unpack2<bf16_t>(u.w, v[6], v[7]);
}
__device__ __forceinline__ void st8(bf16_t *p, const float (&v)[8])
{
    uint4 u;
    u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]); u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
    *reinterpret_cast<uint4 *>(p) = u;
}
__device__ __forceinline__ void ld8(const f16_t *p, float (&v)[8])
{
    const uint4 u = *reinterpret_cast<const uint4 *>(p);
    unpack2<f16_t>(u.x, v[0], v[1]); unpack2<f16_t>(u.y, v[2], v[3]); unpack2<f16_t>(u.z, v[4], v[5]); unpack2<f16_t>(u.w, v[6], v[7]);
}
__device__ __forceinline__ void st8(f16_t *p, const float (&v)[8])
{
    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
    h16x8 h;
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = (_Float16)v[k];
    *reinterpret_cast<h16x8 *>(p) = h;
}

// MFMA 32x32 accumulator (lane = column, half h; register 4q+k = row 8q+4h+k): the two lane halves trade registers
// (v_permlane32_swap) so that afterwards registers 8p..8p+7 of a lane are the 8 CONSECUTIVE rows 16p+8h+{0..7}.
// The epilogues then move 8 channels per access instead of 4: half the memory instructions, half the partial lines.
typedef float dcf_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void acc_rows8(dcf_f32x16 &c)
{
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // upper lanes of the first operand (rows 16p+4+k) <-> lower lanes of the second (rows 16p+8+k)
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[8 * p + k]), __float_as_uint(c[8 * p + 4 + k]), false, false);
            c[8 * p + k] = __uint_as_float(r[0]);
            c[8 * p + 4 + k] = __uint_as_float(r[1]);
        }
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define DCF_DISPATCH_DTYPE(dtype, ...)                        \
    if ((dtype) == DCF_F32) {                                 \
        typedef float T;                                      \
        __VA_ARGS__                                           \
    } else if ((dtype) == DCF_BF16) {                         \
        typedef bf16_t T;                                     \
        __VA_ARGS__                                           \
    } else if ((dtype) == DCF_F16) {                          \
        typedef f16_t T;                                      \
        __VA_ARGS__                                           \
    } else {                                                  \
        dcf_set_error("unsupported dtype %d", (int)(dtype));  \
        return DCF_EUNSUPPORTED;                              \
    }
