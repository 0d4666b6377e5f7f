// conv.hip -- implicit-GEMM convolution on the gfx950 matrix cores (MFMA 32x32).
//
// Replaces every nn.Conv2d (+ folded eval-mode BatchNorm2d + residual add + ReLU) of
// /root/reference/model.py:15-41 (ResidualBlock) and :147-157 (FPN, heads), forward,
// input-gradient and weight-gradient.
//
// Data layout: activations NHWC, i.e. a matrix [pixels][channels] whose rows are
// contiguous channel vectors; weights [Cout][kh][kw][Cin], i.e. [N][K] with K =
// (tap, channel) contiguous -- so the implicit GEMM is the "A row-major x B^T row-major"
// form and both operands stage as rows of contiguous K bytes.
//
//   forward / dgrad:  D[n][m] = sum_{tap,c} Wt[n][tap][c] * X[src(m,tap)][c]
//       m = output pixel, n = output channel; the weight tile is the MFMA A operand so
//       each lane ends up with 4 consecutive output channels of one pixel (8/16 B stores).
//       dgrad is the same kernel with the transposed-conv gather (TRANSPOSED) and the
//       [Cin][tap][Cout] weight image written by dcf_weight_prep.
//   wgrad:            G[co][tap][ci] = sum_p gy[p][co] * x[src(p,tap)][ci]
//       reduction over pixels = the ROW index of both NHWC operands, so fragments are read
//       transposed out of LDS (ds_read_b64_tr_b16 for bf16, ds_read_b32 for fp32); each
//       wave is an independent split-K worker with a wave-private LDS region (no barriers)
//       and writes its own fp32 slab with plain 128-B row stores (fixed-order reduce later).
//
// dtype: bf16 -> v_mfma_f32_32x32x16_bf16;  fp32 -> v_mfma_f32_32x32x2_f32 (exact fp32,
// used by the parity tests against the reference's fp32 CPU path).  fp32 accumulate.
#include <stdlib.h>

#include <algorithm>
#include <vector>
#include <type_traits>
#include "dcf_common.h"
#include "conv_common.h"

namespace {

struct ConvArgs {
    const char *x;       // gathered tensor [B][Hi][Wi][Ck]
    const char *w;       // [Cn][taps][Ck]
    const float *shift;  // [Cn] or null
    const char *res;     // [M][Cn] or null
    char *y;             // [M][Cn]
    int B, Hi, Wi, Ck;
    int Ho, Wo, Cn;
    int kh, kw, stride, pad, relu;
    int M;
    int pixbytes;        // byte pitch between adjacent input pixels (= Ck*esize except for the stem)
    unsigned xbytes, wbytes;  // sizes of the gathered tensor and of the weight image (buffer descriptors)
    const char *mask;         // [M][Cn] or null: output *= (mask > 0)
    const float *rowscale = nullptr;   // [M] or null: the shift enters as rowscale[m] * shift[c] (dcf_conv2d_fwd_rowscale)
    const char *resq = nullptr;   // stride-2 dgrad with parity classes only: residual living on the (2i, 2j) sub-grid of the
                              // output, [B][ceil(Ho/2)][ceil(Wo/2)][Cn] (the input gradient of a 1x1 / stride-2 shortcut)
    // stride-2 dgrad only: output pixels are enumerated parity class by parity class ((oh+pad)&1, (ow+pad)&1),
    // each class padded to whole pixel tiles, so a tile only walks the taps that can hit a real gy pixel
    int parity;               // 1 = class-major enumeration in use
    int cls_tile[5];          // first pixel-tile of each class (prefix sums), in tiles of the launch's BM
    int cls_h[2], cls_w[2];   // rows / columns per parity
    int cls_h0[2], cls_w0[2]; // first row / column of each parity
    int dbg;                  // experiments only (DCF_IGEMM_DBG)
};

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

__device__ long long g_dcf_dbg_t[8];     // DCF_WGRAD3_DBG / DCF_IGEMM_DBG & 2: phase timestamps (s_memtime) of workgroup 0, wave 0
#define DCF_STAMP(i) do { if ((DCF_DBG(a) & 2) && blockIdx.x == 0 && threadIdx.x == 0) g_dcf_dbg_t[i] = clock64(); } while (0)
// DBG & 4: first start / last end over ALL workgroups on the 100 MHz wall clock, plus the sum of workgroup lifetimes
__device__ unsigned long long g_dcf_dbg_w[4];
#define DCF_WSTART() long long w_start__ = 0; do { if ((DCF_DBG(a) & 4) && threadIdx.x == 0) { w_start__ = wall_clock64(); atomicMin(&g_dcf_dbg_w[0], (unsigned long long)w_start__); } } while (0)
#define DCF_WEND() do { if ((DCF_DBG(a) & 4) && threadIdx.x == 0) { const long long e__ = wall_clock64(); atomicMax(&g_dcf_dbg_w[1], (unsigned long long)e__); atomicAdd(&g_dcf_dbg_w[2], (unsigned long long)(e__ - w_start__)); atomicMax(&g_dcf_dbg_w[3], (unsigned long long)(e__ - w_start__)); } } while (0)

// ------------------------------------------------------------------------------------
// forward / dgrad kernel.  Block = 256 threads = WN x WM waves; wave tile TN*32 output
// channels x TM*32 pixels.  K is walked tap by tap in chunks of KB bytes of channels.
// LDS rows are padded by 16 B: the ds_read_b128 fragment reads are then conflict free.
// ------------------------------------------------------------------------------------
template <typename T, int KB, int TN, int TM, int WN, int WM, bool TRANSPOSED, bool DB>
__global__ void __launch_bounds__(256) k_conv_igemm(ConvArgs a)
{
    constexpr int ES = DT<T>::size;
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    constexpr int PITCH = KB + 16;
    constexpr int CPR = KB / 16;                 // 16-byte chunks per row
    constexpr int NCW = BN * CPR, NCX = BM * CPR;
    constexpr int NLW = (NCW + 255) / 256, NLX = (NCX + 255) / 256;
    constexpr int STAGE = (BN + BM) * PITCH;      // one K-chunk of both operands
    // DB: two LDS buffers, one barrier per K-chunk (wins when few workgroups share a CU: small-M layers);
    // !DB: one buffer, two barriers, half the LDS => more resident workgroups (wins on the large layers).
    __shared__ __attribute__((aligned(16))) char lds[(DB ? 2 : 1) * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid / WM, wm = wid % WM;
    const int r = lane & 31, h = lane >> 5;
    DCF_STAMP(0);
    DCF_WSTART();
    // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, so XCD x takes
    // the x-th contiguous chunk of the (pixel-tile, channel-tile) list, channel tiles fastest: vertically
    // adjacent pixel tiles (shared halo rows) and the channel tiles of one pixel tile (same input rows)
    // hit the same L2 instead of each pulling their own copy over the fabric.
    const int nt = a.Cn / BN;
    const bool PAR = TRANSPOSED && a.parity;
    const int mtiles = PAR ? a.cls_tile[4] : cdiv_dev(a.M, BM);
    const int nblk = mtiles * nt;
    const int chunk = (nblk + 7) >> 3;
    const int gidx = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (gidx >= nblk) return;
    const int n0 = (gidx % nt) * BN;
    const int mt = gidx / nt;
    int m0 = mt * BM;                              // first (class-local when PAR) pixel of this tile
    int ph = 0, pw = 0, clsM = a.M;               // parity class of the tile, pixels in the class
    if (PAR) {
        // The four classes of one image region sit next to each other in the tile list, i.e. on the same XCD at about
        // the same time: the half-line stores of horizontally adjacent pixels (a pixel narrower than a 128-B line
        // belongs to a different class than its neighbour) then meet in that XCD's L2 instead of reaching memory
        // as two partial writes from two L2s.
        const int cls = mt & 3;
        ph = cls >> 1; pw = cls & 1;
        m0 = (mt >> 2) * BM;
        clsM = a.B * a.cls_h[ph] * a.cls_w[pw];
        if (m0 >= clsM) return;
    }
    // class-local pixel index -> linear output pixel (identity without parity classes), -1 past the end
    auto out_pixel = [&](int m) -> int {
        if (m >= clsM) return -1;
        if (!PAR) return m;
        const int hwc = a.cls_h[ph] * a.cls_w[pw];
        const int b = m / hwc;
        const int rem = m - b * hwc;
        const int i = rem / a.cls_w[pw], j = rem - i * a.cls_w[pw];
        return (b * a.Ho + a.cls_h0[ph] + 2 * i) * a.Wo + a.cls_w0[pw] + 2 * j;
    };
    const int taps = a.kh * a.kw;
    const int rowbytes = a.Ck * ES;              // bytes of one pixel's channel vector
    const int cchunks = rowbytes / KB;

    // Buffer descriptors: an offset past the end reads as zero, so padding / tile tails need no branches
    // and all addressing is 32-bit (one VALU add per load in the K loop).
    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcW = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;

    // per-thread gather bookkeeping for its X chunks (fixed over the K loop)
    int xb[NLX], xh[NLX], xw[NLX];
    unsigned xoff[NLX], woff[NLW];
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
        const int c = tid + i * 256;
        const int row = c / CPR;
        const int m = (c < NCX) ? out_pixel(m0 + row) : -1;
        xoff[i] = (c % CPR) * 16;
        if (m >= 0) {
            const int b = m / (a.Ho * a.Wo);
            const int rem = m - b * (a.Ho * a.Wo);
            const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            xb[i] = b * a.Hi * a.Wi;
            if (TRANSPOSED) { xh[i] = oh + a.pad; xw[i] = ow + a.pad; }
            else { xh[i] = oh * a.stride - a.pad; xw[i] = ow * a.stride - a.pad; }
        } else {
            xb[i] = -1; xh[i] = 0; xw[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int c = tid + i * 256;
        woff[i] = (NCW % 256 == 0 || c < NCW) ? (unsigned)(n0 + c / CPR) * (unsigned)(taps * rowbytes) + (c % CPR) * 16 : OOB;
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    unsigned pix[NLX];  // byte offset of this tap's source pixel (+ chunk), or OOB
    auto set_tap = [&](int ki, int kj) {
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            int ih, iw;
            bool ok = xb[i] >= 0;
            if (TRANSPOSED) {
                const int th = xh[i] - ki, tw = xw[i] - kj;
                ok = ok && (th >= 0) && (tw >= 0);
                if (a.stride == 2) { ok = ok && !((th | tw) & 1); ih = th >> 1; iw = tw >> 1; }
                else { ih = th; iw = tw; }
                ok = ok && (ih < a.Hi) && (iw < a.Wi);
            } else {
                ih = xh[i] + ki; iw = xw[i] + kj;
                ok = ok && (ih >= 0) && (ih < a.Hi) && (iw >= 0) && (iw < a.Wi);
            }
            pix[i] = ok ? (unsigned)(xb[i] + ih * a.Wi + iw) * (unsigned)a.pixbytes + xoff[i] : OOB;
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

    // taps walked by this tile: all of them, or (stride-2 dgrad) only those of the tile's parity class
    const int tstep = PAR ? 2 : 1;
    const int nki = PAR ? (a.kh - ph + 1) / 2 : a.kh, nkj = PAR ? (a.kw - pw + 1) / 2 : a.kw;
    const int nit = nki * nkj * cchunks;
    int ki = ph, kj = pw, cc = 0;
    auto advance = [&]() {
        if (++cc == cchunks) {
            cc = 0;
            kj += tstep;
            if (kj >= a.kw) { kj = pw; ki += tstep; }
            set_tap(ki, kj);
        }
    };
    auto koff = [&]() { return (unsigned)((ki * a.kw + kj) * cchunks + cc) * KB; };   // K offset of (tap, chunk) in a weight row
    auto compute = [&](int buf) {
        const char *ldsW = lds + buf * STAGE, *ldsX = ldsW + BN * PITCH;
#pragma unroll
        for (int ks = 0; ks < KB / 32; ++ks) {
            uint4 fa[TN], fb[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                fa[i] = *reinterpret_cast<const uint4 *>(ldsW + ((wn * TN + i) * 32 + r) * PITCH + ks * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                fb[j] = *reinterpret_cast<const uint4 *>(ldsX + ((wm * TM + j) * 32 + r) * PITCH + ks * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
    };
    DCF_STAMP(1);
    if (nit > 0) {
        set_tap(ki, kj);
        load_global(koff(), 0);
        store_lds(0);
    }
    DCF_STAMP(2);
    if constexpr (DB) {
        // while the MFMAs of chunk `it` run out of buffer it&1, chunk it+1 (already in registers) is written to
        // the other buffer and chunk it+2 is requested from L2
        if (nit > 1) { advance(); load_global(koff(), (unsigned)cc * KB); }
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            compute(it & 1);
            if (it + 1 < nit) {
                store_lds((it + 1) & 1);              // buffer (it+1)&1 was last read in iteration it-1
                if (it + 2 < nit) { advance(); load_global(koff(), (unsigned)cc * KB); }
            }
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            const bool more = (it + 1 < nit);
            if (more) { advance(); load_global(koff(), (unsigned)cc * KB); }   // flies under the MFMAs
            compute(0);
            __syncthreads();
            if (more) {
                store_lds(0);
                __syncthreads();
            }
        }
    }

    DCF_STAMP(3);
    // epilogue: the MFMA leaves lane (pixel r, half h) with channels 8q+4h+{0..3} of each 32-channel tile.
    //   v = acc + shift + res ; relu ; (dgrad only) v *= (mask > 0), i.e. the ReLU backward of the
    //   layer that PRODUCED this tensor (its dbeta sums come out of the wgrad kernel).
    T *y = reinterpret_cast<T *>(a.y);
    const T *res = reinterpret_cast<const T *>(a.res);
    const T *mask = reinterpret_cast<const T *>(a.mask);
    // after acc_rows8 lane (pixel r, half h) holds channels 16p+8h+{0..7} of each 32-channel tile in registers 8p..8p+7
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc_rows8(acc[i][j]);
    int mpix[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) mpix[j] = out_pixel(m0 + (wm * TM + j) * 32 + r);
    if (PAR && a.resq && ph == (a.pad & 1) && pw == (a.pad & 1)) {
        // half-resolution residual: the class whose rows / columns start at 0 is the (2i, 2j) sub-grid, and the class-local
        // pixel index is that tensor's own linear index
        int mq[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) { const int m = m0 + (wm * TM + j) * 32 + r; mq[j] = m < clsM ? m : -1; }
        conv_epilogue_add<T, TN, TM>(acc, mq, n0 + wn * TN * 32 + 8 * h, a.Cn, reinterpret_cast<const T *>(a.resq));
    }
    conv_epilogue_phases<T, TN, TM>(acc, mpix, n0 + wn * TN * 32 + 8 * h, a.Cn, a.shift, res, mask, a.relu, y, a.rowscale);
    DCF_STAMP(4);
    DCF_WEND();
}

// ------------------------------------------------------------------------------------
// forward / dgrad kernel with LDS-DMA staging (bf16, K chunks of 128 B).  Same tiling, tap walk, XCD-aware tile order,
// parity classes and epilogue as k_conv_igemm; what changes is how a K chunk reaches LDS: `buffer_load ... lds` pieces
// (8 rows x 128 B, out-of-range rows = zero padding) fill an NS-deep ring NS-1 chunks ahead of the MFMAs, so the
// dependent chain load -> ds_write -> barrier -> ds_read of the register-staged kernel (about 900 clocks per chunk on the
// small-M layers, against 128 clocks of MFMA) is off the critical path.  One raw barrier per chunk:
//     wait own pieces of chunk i (counted vmcnt) -> s_barrier -> issue chunk i+NS-1 into the slot chunk i-1 just left -> MFMAs.
// LDS rows have no pad (the DMA image is lane-linear): 16-B chunk c of row R sits at position c ^ ((R >> 1) & 7), which
// keeps the 16 rows of a ds_read_b128 lane group on distinct banks; the swizzle is applied to the source chunk and to the reads.
// ------------------------------------------------------------------------------------
template <typename T, int KB, int TN, int TM, int WN, int WM, bool TRANSPOSED, int NS>
__global__ void __launch_bounds__(256) k_conv_igemm_dma(ConvArgs a)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    static_assert(KB == 128 || KB == 64, "K chunks of 128 or 64 bytes");
    constexpr int ES = 2;
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    constexpr int RP = 1024 / KB, CR = KB / 16;        // rows per 1-KB DMA piece, 16-byte chunks per row
    constexpr int PW = BN / RP / 4, PX = BM / RP / 4;  // DMA pieces per wave and chunk: weights, pixels
    static_assert(PW >= 1 && PX >= 1, "tile too small for this chunk size");
    constexpr int PPW = PW + PX;
    constexpr int STAGE = (BN + BM) * KB;
    // swizzle key of a row: KB 128 -> (R >> 1) & 7 (8 chunks per row), KB 64 -> (R >> 2) & 3 (4 chunks per row): the 16
    // rows of a ds_read_b128 lane group then sit on distinct banks
    auto rowkey = [](int R) { return KB == 128 ? ((R >> 1) & 7) : ((R >> 2) & 3); };
    static_assert((NS - 1) * PPW < 64, "vmcnt range");
    __shared__ __attribute__((aligned(1024))) char lds[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid / WM, wm = wid % WM;
    const int r = lane & 31, h = lane >> 5;
    DCF_STAMP(0);
    DCF_WSTART();
    const int nt = a.Cn / BN;
    const bool PAR = TRANSPOSED && a.parity;
    const int mtiles = PAR ? a.cls_tile[4] : cdiv_dev(a.M, BM);
    const int nblk = mtiles * nt;
    const int chunk = (nblk + 7) >> 3;
    const int gidx = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (gidx >= nblk) return;
    const int n0 = (gidx % nt) * BN;
    const int mt = gidx / nt;
    int m0 = mt * BM;
    int ph = 0, pw = 0, clsM = a.M;
    if (PAR) {
        const int cls = mt & 3;
        ph = cls >> 1; pw = cls & 1;
        m0 = (mt >> 2) * BM;
        clsM = a.B * a.cls_h[ph] * a.cls_w[pw];
        if (m0 >= clsM) return;
    }
    auto out_pixel = [&](int m) -> int {
        if (m >= clsM) return -1;
        if (!PAR) return m;
        const int hwc = a.cls_h[ph] * a.cls_w[pw];
        const int b = m / hwc;
        const int rem = m - b * hwc;
        const int i = rem / a.cls_w[pw], j = rem - i * a.cls_w[pw];
        return (b * a.Ho + a.cls_h0[ph] + 2 * i) * a.Wo + a.cls_w0[pw] + 2 * j;
    };
    const int taps = a.kh * a.kw;
    const int rowbytes = a.Ck * ES;
    const int cchunks = rowbytes / KB;
    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcW = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const unsigned lds0 = lds_addr(lds);

    // DMA side: wave w owns weight pieces w*PW.. and pixel pieces w*PX..; lane = (row l8 of the piece, position chunk lc)
    const int l8 = lane / CR, lc = lane % CR;
    unsigned woff[PW];
    int xb[PX], xh[PX], xw[PX];
    unsigned xoff[PX];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int row = (wid * PW + i) * RP + l8;
        woff[i] = (unsigned)(n0 + row) * (unsigned)(taps * rowbytes) + (unsigned)((lc ^ rowkey(row)) * 16);
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int row = (wid * PX + i) * RP + l8;
        const int m = out_pixel(m0 + row);
        xoff[i] = (unsigned)((lc ^ rowkey(row)) * 16);
        if (m >= 0) {
            const int b = m / (a.Ho * a.Wo);
            const int rem = m - b * (a.Ho * a.Wo);
            const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            xb[i] = b * a.Hi * a.Wi;
            if (TRANSPOSED) { xh[i] = oh + a.pad; xw[i] = ow + a.pad; }
            else { xh[i] = oh * a.stride - a.pad; xw[i] = ow * a.stride - a.pad; }
        } else {
            xb[i] = -1; xh[i] = 0; xw[i] = 0;
        }
    }
    unsigned pix[PX];
    auto set_tap = [&](int ki, int kj) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            int ih, iw;
            bool ok = xb[i] >= 0;
            if (TRANSPOSED) {
                const int th = xh[i] - ki, tw = xw[i] - kj;
                ok = ok && (th >= 0) && (tw >= 0);
                if (a.stride == 2) { ok = ok && !((th | tw) & 1); ih = th >> 1; iw = tw >> 1; }
                else { ih = th; iw = tw; }
                ok = ok && (ih < a.Hi) && (iw < a.Wi);
            } else {
                ih = xh[i] + ki; iw = xw[i] + kj;
                ok = ok && (ih >= 0) && (ih < a.Hi) && (iw >= 0) && (iw < a.Wi);
            }
            pix[i] = ok ? (unsigned)(xb[i] + ih * a.Wi + iw) * (unsigned)a.pixbytes + xoff[i] : OOB;
        }
    };
    const int tstep = PAR ? 2 : 1;
    const int nki = PAR ? (a.kh - ph + 1) / 2 : a.kh, nkj = PAR ? (a.kw - pw + 1) / 2 : a.kw;
    const int nit = nki * nkj * cchunks;
    int ki = ph, kj = pw, cc = 0;
    auto advance = [&]() {
        if (++cc == cchunks) {
            cc = 0;
            kj += tstep;
            if (kj >= a.kw) { kj = pw; ki += tstep; }
            set_tap(ki, kj);
        }
    };
    auto issue = [&](int slot) {      // chunk (ki, kj, cc) -> ring slot
        const unsigned koff = (unsigned)((ki * a.kw + kj) * cchunks + cc) * KB, ccoff = (unsigned)cc * KB;
        const unsigned dw = __builtin_amdgcn_readfirstlane(lds0 + slot * STAGE + wid * PW * 1024);
        const unsigned dx = __builtin_amdgcn_readfirstlane(lds0 + slot * STAGE + BN * KB + wid * PX * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) glds16(srcW, woff[i] + koff, dw + i * 1024);
#pragma unroll
        for (int i = 0; i < PX; ++i) glds16(srcX, pix[i] == OOB ? OOB : pix[i] + ccoff, dx + i * 1024);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // read side: row r of every 32-row tile has swizzle key (r >> 1) & 7; k-step ks, lane half h wants source chunk 2 ks + h
    const int key = rowkey(r);
    const int rdW = (wn * TN * 32 + r) * KB, rdX = BN * KB + (wm * TM * 32 + r) * KB;
    int swz[KB / 32];
#pragma unroll
    for (int ks = 0; ks < KB / 32; ++ks) swz[ks] = ((2 * ks + h) ^ key) * 16;
    auto compute = [&](int slot) {
        const char *base = lds + slot * STAGE;
#pragma unroll
        for (int ks = 0; ks < KB / 32; ++ks) {
            uint4 fa[TN], fb[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4 *>(base + rdW + i * 32 * KB + swz[ks]);
#pragma unroll
            for (int j = 0; j < TM; ++j) fb[j] = *reinterpret_cast<const uint4 *>(base + rdX + j * 32 * KB + swz[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
    };

    DCF_STAMP(1);
    if (nit > 0) set_tap(ki, kj);
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nit) { if (s0) advance(); issue(s0); }
    DCF_STAMP(2);
    int slot = 0, islot = NS - 1;
    for (int it = 0; it < nit; ++it) {
        const int ahead = min(NS - 2, nit - 1 - it);       // chunks issued after chunk `it` that may stay in flight
        if (ahead >= 2) wait_vmcnt<(NS > 3 ? 2 : 0) * PPW>();
        else if (ahead == 1) wait_vmcnt<(NS > 2 ? 1 : 0) * PPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                       // everyone's pieces of chunk `it` landed; chunk it-1 fully consumed
        if (it + NS - 1 < nit) { advance(); issue(islot); }
        compute(slot);
        slot = slot + 1 == NS ? 0 : slot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }

    DCF_STAMP(3);
    T *y = reinterpret_cast<T *>(a.y);
    const T *res = reinterpret_cast<const T *>(a.res);
    const T *mask = reinterpret_cast<const T *>(a.mask);
    // after acc_rows8 lane (pixel r, half h) holds channels 16p+8h+{0..7} of each 32-channel tile in registers 8p..8p+7
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc_rows8(acc[i][j]);
    int mpix[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) mpix[j] = out_pixel(m0 + (wm * TM + j) * 32 + r);
    if (PAR && a.resq && ph == (a.pad & 1) && pw == (a.pad & 1)) {
        // half-resolution residual: the class whose rows / columns start at 0 is the (2i, 2j) sub-grid, and the class-local
        // pixel index is that tensor's own linear index
        int mq[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) { const int m = m0 + (wm * TM + j) * 32 + r; mq[j] = m < clsM ? m : -1; }
        conv_epilogue_add<T, TN, TM>(acc, mq, n0 + wn * TN * 32 + 8 * h, a.Cn, reinterpret_cast<const T *>(a.resq));
    }
    conv_epilogue_phases<T, TN, TM>(acc, mpix, n0 + wn * TN * 32 + 8 * h, a.Cn, a.shift, res, mask, a.relu, y, a.rowscale);
    DCF_STAMP(4);
    DCF_WEND();
}

template <typename T, bool TR>
int launch_igemm(const ConvArgs &a_in, hipStream_t s, const char *base, double flops)
{
    // algorithmic bytes: the gathered tensor and the weights read once, the output written once (+ residual / mask reads)
    const double bytes = (double)a_in.xbytes + (double)a_in.wbytes + (double)a_in.M * a_in.Cn * DT<T>::size * (1 + (a_in.res ? 1 : 0) + (a_in.mask ? 1 : 0));
    char name[64];
    const ConvArgs &a = a_in;
    constexpr int ES = DT<T>::size;
    const int rowbytes = a.Ck * ES;
    const bool kb128 = (rowbytes % 128) == 0;
#define DCF_IGEMM(KB_, TN_, TM_, WN_, WM_)                                                                          \
    do {                                                                                                            \
        constexpr int BN_ = WN_ * TN_ * 32, BM_ = WM_ * TM_ * 32;                                                   \
        ConvArgs a = a_in;                                                                                          \
                a.dbg = dcf_ablate_opt("IGEMM_DBG");                                                                        \
        int mtiles = cdiv(a.M, BM_);                                                                                \
        if (a.parity) {   /* tile list = (region, class) with the class fastest; every class gets the largest class's count */ \
            int mx = 0;                                                                                             \
            for (int c = 0; c < 4; ++c) mx = std::max(mx, cdiv((int64_t)a.B * a.cls_h[c >> 1] * a.cls_w[c & 1], BM_)); \
            a.cls_tile[4] = mtiles = 4 * mx;                                                                        \
        }                                                                                                           \
        dim3 grid((((int64_t)mtiles * (a.Cn / BN_) + 7) / 8) * 8);                                                  \
        const bool db = (int64_t)mtiles * (a.Cn / BN_) <= 512;   /* <= 2 workgroups per CU: 1-barrier pipeline */   \
        snprintf(name, sizeof(name), "%s<%d,%d,%d,%d,%d%s>", base, KB_, TN_, TM_, WN_, WM_, db ? ",db" : "");        \
        if (DCF_DBG(a) & 4) {                                                                                            \
            unsigned long long w[4] = {~0ull, 0, 0, 0};                                                             \
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dcf_dbg_w), w, sizeof(w));                                               \
        }                                                                                                           \
        constexpr int NS_ = (BN_ + BM_) <= 128 ? 4 : 3;                                                              \
        static DcfOpt dma_env_o("IGEMM_DMA"); const char *dma_env = dma_env_o.str();                                                       \
        const int dma_mode = dma_env ? atoi(dma_env) : 1;      /* 0 off, 1 small-M tiles, 2 every bf16 KB=128 launch */ \
        bool launched = false;                                                                                      \
        if constexpr (ES == 2 && KB_ == 128) {                                                                      \
            if (dma_mode && (dma_mode == 2 || db)) {                                                                 \
                snprintf(name, sizeof(name), "%s<%d,%d,%d,%d,%d,dma%d>", base, KB_, TN_, TM_, WN_, WM_, NS_);        \
                DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv_igemm_dma<T, 128, TN_, TM_, WN_, WM_, TR, NS_>), grid, dim3(256), 0, s, a)); \
                launched = true;                                                                                    \
            } else if (dma_mode == 3) {   /* many workgroups: 2-deep ring keeps 2+ of them per CU */                  \
                snprintf(name, sizeof(name), "%s<%d,%d,%d,%d,%d,dma2>", base, KB_, TN_, TM_, WN_, WM_);              \
                DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv_igemm_dma<T, 128, TN_, TM_, WN_, WM_, TR, 2>), grid, dim3(256), 0, s, a)); \
                launched = true;                                                                                    \
            }                                                                                                       \
        }                                                                                                           \
        if (launched) {                                                                                             \
        } else if (db) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv_igemm<T, KB_, TN_, TM_, WN_, WM_, TR, true>), grid, dim3(256), 0, s, a)); \
        else DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv_igemm<T, KB_, TN_, TM_, WN_, WM_, TR, false>), grid, dim3(256), 0, s, a)); \
        if (DCF_DBG(a) & 4) {                                                                                            \
            unsigned long long w[4];                                                                                \
            (void)hipStreamSynchronize(s);                                                                              \
            (void)hipMemcpyFromSymbol(w, HIP_SYMBOL(g_dcf_dbg_w), sizeof(w));                                             \
            fprintf(stderr, "[%s M=%d Ck=%d Cn=%d taps=%d blocks=%d] span %.2f us, mean workgroup life %.2f us, max %.2f us\n", name, a.M, a.Ck, a.Cn, a.kh * a.kw, (int)grid.x, \
                    (w[1] - w[0]) * 0.01, w[2] * 0.01 / grid.x, w[3] * 0.01);                                       \
        }                                                                                                           \
        if (DCF_DBG(a) & 2) {                                                                                            \
            long long tt[8];                                                                                        \
            (void)hipStreamSynchronize(s);                                                                              \
            (void)hipMemcpyFromSymbol(tt, HIP_SYMBOL(g_dcf_dbg_t), sizeof(tt));                                           \
            fprintf(stderr, "[%s M=%d Ck=%d Cn=%d taps=%d blocks=%d] clocks: setup %lld first-stage %lld loop %lld epilogue %lld\n", name, a.M, a.Ck, a.Cn, a.kh * a.kw, (int)grid.x, \
                    tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3]);                                    \
        }                                                                                                           \
        return DCF_OK;                                                                                              \
    } while (0)
    // tile choice: the biggest tile that still gives the chip >= ~1 workgroup per CU
    // (DCF_TILE=0..3 forces a tile for experiments: 128x128, 64x128, 64x64, 32x128 channels x pixels)
    static DcfOpt force_env_o("TILE"); const char *force_env = force_env_o.str();
    const int force = force_env ? atoi(force_env) : -1;
    if (force == 0 && a.Cn % 128 == 0) { if (kb128) DCF_IGEMM(128, 2, 2, 2, 2); else DCF_IGEMM(64, 2, 2, 2, 2); }
    if (force == 1 && a.Cn % 64 == 0) { if (kb128) DCF_IGEMM(128, 2, 1, 1, 4); else DCF_IGEMM(64, 2, 1, 1, 4); }
    if (force == 2 && a.Cn % 64 == 0) { if (kb128) DCF_IGEMM(128, 1, 1, 2, 2); else DCF_IGEMM(64, 1, 1, 2, 2); }
    if (force == 3) { if (kb128) DCF_IGEMM(128, 1, 1, 1, 4); else DCF_IGEMM(64, 1, 1, 1, 4); }
    if (force == 4 && a.Cn % 128 == 0) { if (kb128) DCF_IGEMM(128, 2, 1, 2, 2); else DCF_IGEMM(64, 2, 1, 2, 2); }      // 128 ch x 64 px
    const int64_t want_blocks = 256;
    auto blocks = [&](int bn, int bm) { return (int64_t)cdiv(a.M, bm) * (a.Cn / bn); };
    if (a.Cn % 128 == 0 && blocks(128, 128) >= want_blocks) {
        if (kb128) DCF_IGEMM(128, 2, 2, 2, 2); else DCF_IGEMM(64, 2, 2, 2, 2);
    } else if (a.Cn % 64 == 0 && blocks(64, 128) >= want_blocks) {
        if (kb128) DCF_IGEMM(128, 2, 1, 1, 4); else DCF_IGEMM(64, 2, 1, 1, 4);
    } else if (a.Cn % 64 == 0) {          // small M: 64x64 tiles give the most workgroups
        if (kb128) DCF_IGEMM(128, 1, 1, 2, 2); else DCF_IGEMM(64, 1, 1, 2, 2);
    } else {
        if (kb128) DCF_IGEMM(128, 1, 1, 1, 4); else DCF_IGEMM(64, 1, 1, 1, 4);
    }
#undef DCF_IGEMM
}

// ------------------------------------------------------------------------------------
// wgrad kernel.  grid.x = co_tiles*ci_tiles*taps, grid.y = nsplit/4; every wave is its
// own split-K worker over a contiguous pixel range, staging PK pixels per step into a
// wave-private LDS region ([pixel][channel] rows, read back transposed).
// ------------------------------------------------------------------------------------
struct WgArgs {
    const char *x;   // [B][H][W][Cin]
    const char *gy;  // [B][Ho][Wo][Cout]
    float *slabs;    // [nsplit][Cout][taps][Cin]
    float *gsum;     // [nsplit][Cout] or null: per-split column sums of gy (= dL/dbeta of the folded BN)
    int B, H, W, Cin, Ho, Wo, Cout;
    int kh, kw, stride, pad;
    int M, nsplit, per_split;  // per_split = pixels per split (multiple of PK)
    int co_tiles, ci_tiles;
    int pixbytes;              // byte pitch between adjacent x pixels (= Cin*esize except for the stem)
    unsigned xbytes, gbytes;   // tensor sizes for the buffer descriptors
    int dbg;                   // experiments only (DCF_WGRAD3_DBG): 1 = every DMA reads the zero page
    int xcd;                   // generic kernel: 1 = XCD-aware work mapping (wgrad_xcd(): nsplit a multiple of 8, enough ranges);
                               // k_conv_wgrad3g: the XCD the layer's first run of units goes to (0 in a single launch)
    int upx;                   // k_conv_wgrad3g: units per XCD (wgrad_upx())
};

template <typename T, int TM, int TN>
__device__ __forceinline__ void wgrad_body(const WgArgs &a, const int bid)
{
    constexpr int ES = DT<T>::size;
    constexpr int PK = 32;                        // pixels per stage
    constexpr int RA = TM * 32 * ES, RB = TN * 32 * ES;     // row bytes of the two tiles
    // bf16: pitch = 64 or 192 (mod 256) keeps the 4 rows x 64 B of one transposed read on
    // distinct bank groups; fp32 reads are one row per 32-lane half (any pitch is conflict free)
    constexpr int PA = (ES == 2) ? (RA == 64 ? 64 : 192) : RA + 16;
    constexpr int PB = (ES == 2) ? (RB == 64 ? 64 : 192) : RB + 16;
    constexpr int WAVE_LDS = PK * (PA + PB);
    __shared__ __attribute__((aligned(16))) char lds_all[4 * WAVE_LDS];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    char *ldsA = lds_all + wid * WAVE_LDS;
    char *ldsB = ldsA + PK * PA;

    // XCD-aware work mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so all
    // (tile, tap) workgroups of one pixel range get the same (linear id % 8): they re-read the same gy / x
    // pixels 9*tiles times, and now do it out of ONE XCD's L2 instead of the fabric.
    // (Used when the number of pixel ranges is a multiple of 8; otherwise plain tile-major order.)
    const int ninner = a.co_tiles * a.ci_tiles * a.kh * a.kw;
    const int L = bid;
    const bool xcd = a.xcd != 0;                  // (host: wgrad_xcd())
    const int slab_id = xcd ? (L / (8 * ninner)) * 8 + (L & 7) : L / ninner;
    if (slab_id >= a.nsplit) return;          // grouped launches round a layer's grid up to a multiple of 8 workgroups
    int t = xcd ? (L >> 3) % ninner : L % ninner;
    const int tap = t % (a.kh * a.kw); t /= (a.kh * a.kw);
    const int cit = t % a.ci_tiles;
    const int cot = t / a.ci_tiles;
    const int co0 = cot * TM * 32, ci0 = cit * TN * 32;
    const int ki = tap / a.kw, kj = tap - ki * a.kw;
    const int split = slab_id * 4 + wid;
    const int p_begin = split * a.per_split;
    const int p_end = min(p_begin + a.per_split, a.M);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // The waves of (tap 0, ci-tile 0) also accumulate sum_p gy[p][co] from the gy fragments they hold
    // anyway (a few VALU adds, no extra accumulator tile): dbeta falls out of the weight-gradient
    // kernel, per split, bit-reproducibly.
    const bool do_sum = (a.gsum != nullptr) && (tap == 0) && (cit == 0);
    float fsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) fsum[i] = 0.f;

    constexpr int CA = RA / 16, CB = RB / 16;       // 16-B chunks per row
    constexpr int NLA = PK * CA / 64, NLB = PK * CB / 64;
    const int rowA = a.Cout * ES;
    const int coutA = min(TM * 32, a.Cout - co0) * ES;  // valid bytes of this tile's rows
    const int cinB = min(TN * 32, a.Cin - ci0) * ES;

    // Buffer descriptors + per-lane row state.  Lane l stages HALF of pixel row (l >> 1) of both tiles
    // (NLA resp. NLB consecutive 16-byte chunks), so a lane carries one pixel coordinate and one byte
    // offset per tile: the stage loop has no integer division, no 64-bit arithmetic and a handful of
    // VALU ops per load.  Rows past the end / padding taps read as zero through the range check.
    const __amdgpu_buffer_rsrc_t srcG = __builtin_amdgcn_make_buffer_rsrc((void *)a.gy, 0, a.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    static_assert(NLA * 2 == CA && NLB * 2 == CB, "half a row per lane");
    const int lrow = lane >> 1, lhalf = lane & 1;
    int pcur = p_begin + lrow;                       // this lane's pixel of the current stage
    unsigned voA = (unsigned)pcur * (unsigned)rowA + (unsigned)(co0 * ES + lhalf * NLA * 16);
    int bB, ohB, owB;
    {
        const int b = pcur / (a.Ho * a.Wo);
        const int rem = pcur - b * (a.Ho * a.Wo);
        bB = b; ohB = rem / a.Wo; owB = rem - ohB * a.Wo;
    }
    const unsigned colB = (unsigned)(ci0 * ES + lhalf * NLB * 16);
    const int tapH = ki - a.pad, tapW = kj - a.pad;

    uint4 ra[NLA], rb[NLB];
    auto load_stage = [&]() {
        const bool live = pcur < p_end;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const bool ok = live && ((lhalf * NLA + i) * 16 < coutA);
            ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srcG, ok ? voA + i * 16 : OOB, 0, 0));
        }
        const int ih = ohB * a.stride + tapH, iw = owB * a.stride + tapW;
        const bool inimg = live && (ih >= 0) && (ih < a.H) && (iw >= 0) && (iw < a.W);
        const unsigned voB = (unsigned)((bB * a.H + ih) * a.W + iw) * (unsigned)a.pixbytes + colB;
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const bool ok = inimg && ((lhalf * NLB + i) * 16 < cinB);
            rb[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srcX, ok ? voB + i * 16 : OOB, 0, 0));
        }
        pcur += PK;
        voA += (unsigned)(PK * rowA);
        owB += PK;
        while (owB >= a.Wo) { owB -= a.Wo; ++ohB; }
        while (ohB >= a.Ho) { ohB -= a.Ho; ++bB; }
    };
    if (p_begin < p_end) load_stage();
    for (int p0 = p_begin; p0 < p_end; p0 += PK) {
        // previous stage's LDS reads have all been consumed by MFMAs issued before this
        // point in program order of the same wave; ds ops of one wave execute in order.
#pragma unroll
        for (int i = 0; i < NLA; ++i) *reinterpret_cast<uint4 *>(ldsA + lrow * PA + (lhalf * NLA + i) * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < NLB; ++i) *reinterpret_cast<uint4 *>(ldsB + lrow * PB + (lhalf * NLB + i) * 16) = rb[i];
        __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order: no s_barrier needed
        if (p0 + PK < p_end) load_stage();  // next stage's global loads fly under this stage's MFMAs
        if constexpr (ES == 2) {
            // bf16: per K=16 step two transposed 4x16 reads per 32-channel fragment.
            // lane = 16g+4q+p: rows (pixels) kbase+8h+{q, 4+q}, columns 16(g&1)+4p..+3, h = g>>1.
            const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g >> 1;
#pragma unroll
            for (int ks = 0; ks < PK / 16; ++ks) {
                uint4 fa[TM], fb[TN];
                const int row0 = ks * 16 + 8 * hh + q;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const char *base = ldsA + row0 * PA + (i * 32 + 16 * (g & 1) + 4 * pp) * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * PA));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const char *base = ldsB + row0 * PB + (j * 32 + 16 * (g & 1) + 4 * pp) * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * PB));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        Mma<T>::run(fa[i], fb[j], acc[i][j]);
                if (do_sum) {   // lane (channel r, half h) holds 8 of the 16 pixels of this step
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const unsigned w[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { float lo, hi; unpack2<T>(w[e], lo, hi); fsum[i] += lo + hi; }
                    }
                }
            }
        } else {
            // fp32: lane (i = lane&31, k = lane>>5) reads element [pixel 2*ks+k][channel i]
            const int r = lane & 31, kk = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < PK / 2; ++ks) {
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float *>(ldsA + (2 * ks + kk) * PA + (i * 32 + r) * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float *>(ldsB + (2 * ks + kk) * PB + (j * 32 + r) * 4);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
                if (do_sum) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fsum[i] += fa[i];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (do_sum) {   // lanes r and r+32 hold the two pixel halves of channel r
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float tot = fsum[i] + __shfl_xor(fsum[i], 32, 64);
            const int co = co0 + i * 32 + (lane & 31);
            if (lane < 32 && co < a.Cout) a.gsum[(size_t)split * a.Cout + co] = tot;
        }
    }
    // Block-level reduction of the 4 waves' partial tiles through LDS in a fixed order
    // ((w0 + w2) + (w1 + w3)): one slab per BLOCK, so 4x more waves hide latency per slab byte.
    {
        float *red = reinterpret_cast<float *>(lds_all);
        constexpr int TILE = TM * TN * 16 * 64;   // floats of one wave's accumulator tile
        static_assert(2 * TILE * 4 <= 4 * WAVE_LDS, "two accumulator tiles must fit the block's LDS");
        __syncthreads();                          // every wave is done with its staging region
        if (wid >= 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) red[(wid - 2) * TILE + ((i * TN + j) * 16 + q) * 64 + lane] = acc[i][j][q];
        }
        __syncthreads();
        if (wid < 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] += red[wid * TILE + ((i * TN + j) * 16 + q) * 64 + lane];
        }
        __syncthreads();
        if (wid == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) red[((i * TN + j) * 16 + q) * 64 + lane] = acc[i][j][q];
        }
        __syncthreads();
        if (wid != 0) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] += red[((i * TN + j) * 16 + q) * 64 + lane];
    }
    // slab store: acc lane l: column (ci) = l&31, rows (co) = (reg&3)+8(reg>>2)+4(l>>5)
    const int taps = a.kh * a.kw;
    float *slab = a.slabs + (size_t)slab_id * a.Cout * taps * a.Cin;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = ci0 + j * 32 + r;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (co < a.Cout && ci < a.Cin) slab[((size_t)co * taps + tap) * a.Cin + ci] = acc[i][j][q];
            }
        }
}

template <typename T, int TM, int TN>
__global__ void __launch_bounds__(256) k_conv_wgrad(WgArgs a)
{
    wgrad_body<T, TM, TN>(a, blockIdx.x);
}

// ------------------------------------------------------------------------------------
// wgrad kernel for 3x3 / stride 1 / pad 1 layers (all but a handful of the network's convs).
// The generic kernel above stages gy and x once PER TAP (9x each) and moves two LDS fragments per
// MFMA; here one wave owns the three horizontal taps of KR kernel rows and walks the PADDED image
// (rows of Wo + 2 positions; positions 0 and Wo+1 carry gy = 0 and x = 0): in that flattened space
// the x fragment of tap kj is the same staged tile read kj rows further down, with no wrap between
// image rows, so a stage of 32 positions costs one gy tile + one (32+2)-row x tile per kernel row
// for 3*KR taps: a third of the L1 traffic and half the LDS traffic per MFMA.
// grid.x = co_tiles*ci_tiles*(3/KR)*nsplit; 4 waves = 4 position ranges, reduced in fixed order.
// ------------------------------------------------------------------------------------
template <typename T, int TM, int TN, int KR>
__global__ void __launch_bounds__(256, 2) k_conv_wgrad3(WgArgs a)
{
    constexpr int ES = DT<T>::size;
    constexpr int PK = 32, XR = PK + 2;
    constexpr int RA = TM * 32 * ES, RB = TN * 32 * ES;
    constexpr int PA = (ES == 2) ? (RA == 64 ? 64 : 192) : RA + 16;
    constexpr int PB = (ES == 2) ? (RB == 64 ? 64 : 192) : RB + 16;
    constexpr int WAVE_LDS = PK * PA + KR * XR * PB;
    constexpr int TILE = TM * TN * 16 * 64;       // floats of one tap's accumulator tile
    constexpr int LDS_BYTES = 4 * WAVE_LDS > 2 * TILE * 4 ? 4 * WAVE_LDS : 2 * TILE * 4;
    __shared__ __attribute__((aligned(16))) char lds_all[LDS_BYTES];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    char *ldsA = lds_all + wid * WAVE_LDS;
    char *ldsB = ldsA + PK * PA;                  // KR buffers of XR rows

    constexpr int NKG = 3 / KR;
    const int ninner = a.co_tiles * a.ci_tiles * NKG;
    const int L = blockIdx.x;
    const bool xcd = ((a.nsplit & 7) == 0) && (a.nsplit >= 48);
    const int slab_id = xcd ? (L / (8 * ninner)) * 8 + (L & 7) : L / ninner;
    int t = xcd ? (L >> 3) % ninner : L % ninner;
    const int kg = t % NKG; t /= NKG;
    const int cit = t % a.ci_tiles;
    const int cot = t / a.ci_tiles;
    const int co0 = cot * TM * 32, ci0 = cit * TN * 32, ki0 = kg * KR;
    const int Wp = a.Wo + 2;
    const int split = slab_id * 4 + wid;
    const int q_begin = split * a.per_split;
    const int q_end = min(q_begin + a.per_split, a.M);     // a.M = B*Ho*(Wo+2) padded positions

    f32x16 acc[KR][3][TM][TN];
#pragma unroll
    for (int r = 0; r < KR; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[r][k][i][j][q] = 0.f;

    const bool do_sum = (a.gsum != nullptr) && (kg == 0) && (cit == 0);
    float fsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) fsum[i] = 0.f;

    constexpr int CA = RA / 16, CB = RB / 16;
    constexpr int NLA = PK * CA / 64, NLB = PK * CB / 64;
    static_assert(NLA * 2 == CA && NLB * 2 == CB, "half a row per lane");
    const int rowA = a.Cout * ES;
    const int coutA = min(TM * 32, a.Cout - co0) * ES;
    const int cinB = min(TN * 32, a.Cin - ci0) * ES;
    const __amdgpu_buffer_rsrc_t srcG = __builtin_amdgcn_make_buffer_rsrc((void *)a.gy, 0, a.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int lrow = lane >> 1, lhalf = lane & 1;
    const unsigned colA = (unsigned)(co0 * ES + lhalf * NLA * 16);
    const unsigned colB = (unsigned)(ci0 * ES + lhalf * NLB * 16);

    // lane state: gy position q_begin + lrow -> (image row Rg over the whole batch, padded column cg);
    // x position one further (slots 2..33 of the x tile are positions q0+1 .. q0+32)
    int qg = q_begin + lrow;
    int Rg = qg / Wp, cg = qg - Rg * Wp;
    int Rx = (qg + 1) / Wp, cx = (qg + 1) - Rx * Wp;
    int ohx = Rx % a.Ho;

    auto x_load = [&](int R, int c, int oh, int kr, unsigned col, int nchunk, int chunk0, uint4 *dst) {
        const int ih = oh + ki0 + kr - 1;
        const bool ok = (c >= 1) && (c <= a.Wo) && (ih >= 0) && (ih < a.H);
        const unsigned vo = (unsigned)((R + ki0 + kr - 1) * a.W + c - 1) * (unsigned)a.pixbytes + col;
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const bool okc = ok && (i < nchunk) && ((chunk0 + i) * 16 < cinB);
            dst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srcX, okc ? vo + i * 16 : OOB, 0, 0));
        }
    };

    // prologue: x slots 0 and 1 (positions q_begin-1, q_begin) -- later stages inherit them from the previous stage
    if (q_begin < q_end) {
        const int slot = lane >> 1;
        if (slot < 2) {
            const int qq = q_begin - 1 + slot;
            const int R = qq >= 0 ? qq / Wp : 0, c = qq >= 0 ? qq - R * Wp : 0;   // c = 0 reads as zero
            const int oh = R % a.Ho;
#pragma unroll
            for (int kr = 0; kr < KR; ++kr) {
                uint4 v[NLB];
                x_load(R, c, oh, kr, colB, NLB, lhalf * NLB, v);
#pragma unroll
                for (int i = 0; i < NLB; ++i) *reinterpret_cast<uint4 *>(ldsB + (kr * XR + slot) * PB + (lhalf * NLB + i) * 16) = v[i];
            }
        }
    }

    uint4 ra[NLA], rb[KR][NLB];
    auto load_stage = [&]() {
        const bool live = (qg < q_end) && (cg >= 1) && (cg <= a.Wo);
        const unsigned voA = (unsigned)(Rg * a.Wo + cg - 1) * (unsigned)rowA + colA;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const bool ok = live && ((lhalf * NLA + i) * 16 < coutA);
            ra[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(srcG, ok ? voA + i * 16 : OOB, 0, 0));
        }
#pragma unroll
        for (int kr = 0; kr < KR; ++kr) x_load(Rx, cx, ohx, kr, colB, NLB, lhalf * NLB, rb[kr]);
        qg += PK;
        cg += PK;
        while (cg >= Wp) { cg -= Wp; ++Rg; }
        cx += PK;
        while (cx >= Wp) { cx -= Wp; ++Rx; ++ohx; if (ohx == a.Ho) ohx = 0; }
    };
    if (q_begin < q_end) load_stage();
    for (int q0 = q_begin; q0 < q_end; q0 += PK) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) *reinterpret_cast<uint4 *>(ldsA + lrow * PA + (lhalf * NLA + i) * 16) = ra[i];
#pragma unroll
        for (int kr = 0; kr < KR; ++kr)
#pragma unroll
            for (int i = 0; i < NLB; ++i) *reinterpret_cast<uint4 *>(ldsB + (kr * XR + 2 + lrow) * PB + (lhalf * NLB + i) * 16) = rb[kr][i];
        __builtin_amdgcn_wave_barrier();
        if (q0 + PK < q_end) load_stage();
        if constexpr (ES == 2) {
            const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g >> 1;
#pragma unroll
            for (int ks = 0; ks < PK / 16; ++ks) {
                uint4 fa[TM];
                const int row0 = ks * 16 + 8 * hh + q;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const char *base = ldsA + row0 * PA + (i * 32 + 16 * (g & 1) + 4 * pp) * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * PA));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int kr = 0; kr < KR; ++kr)
#pragma unroll
                    for (int kj = 0; kj < 3; ++kj) {
                        uint4 fb[TN];
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const char *base = ldsB + (kr * XR + row0 + kj) * PB + (j * 32 + 16 * (g & 1) + 4 * pp) * 2;
                            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * PB));
                            uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                            fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                        }
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                Mma<T>::run(fa[i], fb[j], acc[kr][kj][i][j]);
                    }
                if (do_sum) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const unsigned w[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { float lo, hi; unpack2<T>(w[e], lo, hi); fsum[i] += lo + hi; }
                    }
                }
            }
        } else {
            const int r = lane & 31, kk = lane >> 5;
#pragma unroll 4
            for (int ks = 0; ks < PK / 2; ++ks) {
                float fa[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float *>(ldsA + (2 * ks + kk) * PA + (i * 32 + r) * 4);
#pragma unroll
                for (int kr = 0; kr < KR; ++kr)
#pragma unroll
                    for (int kj = 0; kj < 3; ++kj) {
                        float fb[TN];
#pragma unroll
                        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float *>(ldsB + (kr * XR + 2 * ks + kk + kj) * PB + (j * 32 + r) * 4);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) acc[kr][kj][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[kr][kj][i][j], 0, 0, 0);
                    }
                if (do_sum) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fsum[i] += fa[i];
                }
            }
        }
        // the last two x rows of this stage are the first two of the next (same wave, LDS ops in order)
        if (lane < 2 * CB) {
            const int row = lane / CB, ch = lane - row * CB;
#pragma unroll
            for (int kr = 0; kr < KR; ++kr) {
                const uint4 v = *reinterpret_cast<const uint4 *>(ldsB + (kr * XR + PK + row) * PB + ch * 16);
                *reinterpret_cast<uint4 *>(ldsB + (kr * XR + row) * PB + ch * 16) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (do_sum) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float tot = fsum[i] + __shfl_xor(fsum[i], 32, 64);
            const int co = co0 + i * 32 + (lane & 31);
            if (lane < 32 && co < a.Cout) a.gsum[(size_t)split * a.Cout + co] = tot;
        }
    }
    // fixed-order reduction of the 4 waves' tiles, one tap at a time: ((w0 + w2) + (w1 + w3))
    float *red = reinterpret_cast<float *>(lds_all);
    float *slab = a.slabs + (size_t)slab_id * a.Cout * 9 * a.Cin;
    const int r = lane & 31, h = lane >> 5;
    __syncthreads();
#pragma unroll
    for (int kr = 0; kr < KR; ++kr)
#pragma unroll
        for (int kj = 0; kj < 3; ++kj) {
            if (wid >= 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) red[(wid - 2) * TILE + ((i * TN + j) * 16 + q) * 64 + lane] = acc[kr][kj][i][j][q];
            }
            __syncthreads();
            if (wid < 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc[kr][kj][i][j][q] += red[wid * TILE + ((i * TN + j) * 16 + q) * 64 + lane];
            }
            __syncthreads();
            if (wid == 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) red[((i * TN + j) * 16 + q) * 64 + lane] = acc[kr][kj][i][j][q];
            }
            __syncthreads();
            if (wid == 0) {
                const int tap = (ki0 + kr) * 3 + kj;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int ci = ci0 + j * 32 + r;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int co = co0 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                            const float v = acc[kr][kj][i][j][q] + red[((i * TN + j) * 16 + q) * 64 + lane];
                            if (co < a.Cout && ci < a.Cin) slab[((size_t)co * 9 + tap) * a.Cin + ci] = v;
                        }
                    }
            }
            __syncthreads();
        }
}

// ------------------------------------------------------------------------------------
// k_conv_wgrad3g: the 3x3 / stride-1 row-sharing wgrad (see k_conv_wgrad3) with LDS-DMA staging.
// The register-staged kernels above keep ONE stage of loads in flight per wave and pay ~2 us of
// memory latency per 32-pixel stage; here a wave owns an NS-deep ring of LDS slots that
// `global_load_lds_dwordx4` fills directly (no staging VGPRs, no ds_write), NS-1 stages ahead,
// retired by a counted `s_waitcnt vmcnt` (the issuing wave's own wait orders its ds_reads; the
// rings are wave-private, so the main loop has no barrier).  One workgroup per CU (LDS-bound),
// one wave per SIMD, the whole register file for the 12 accumulator tiles.
//   * DMA image is lane-linear (1 KiB = 8 rows x 128 B per instruction), so rows have no pad;
//     128-B rows are XOR-swizzled (64-B half ^= row bit 1) on the SOURCE address and on the
//     transposed reads, which keeps the 4 rows of a ds_read_b64_tr_b16 group on distinct banks.
//   * padding, junk rows and masked channels are out-of-range buffer offsets: the DMA writes zeros for them.
// bf16 only; Wo + 2 >= the rows one stage loads (40 or 48).
// ------------------------------------------------------------------------------------


template <typename T, int TM, int TN, int NS, int NW>
__device__ __forceinline__ void wgrad3g_body(const WgArgs &a, const int bid)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    constexpr int PK = 32, XROWS = PK + 2;
    constexpr int RA = TM * 64, RB = TN * 64;        // row bytes = LDS pitch
    constexpr int LPA = RA / 16, LPB = RB / 16;      // lanes (16-B chunks) per row
    constexpr int RPA = 64 / LPA, RPB = 64 / LPB;    // rows per DMA instruction
    constexpr int NA = PK / RPA;
    constexpr int NB = (XROWS + RPB - 1) / RPB;
    constexpr int GI = NA + NB;                      // DMA instructions per stage
    constexpr int SA = PK * RA, SB = NB * 1024;      // slot bytes
    constexpr int WAVE_LDS = NS * (SA + SB);
    constexpr int TILE = TM * TN * 16 * 64;
    constexpr int RED_BYTES = NW * TILE * 4;         // one tap's tile of every wave
    constexpr int LDS_BYTES = NW * WAVE_LDS > RED_BYTES ? NW * WAVE_LDS : RED_BYTES;
    static_assert((NS - 1) * GI < 64, "vmcnt range");
    static_assert(NS >= 2 && NS <= 4 && (NW == 4 || NW == 8), "ring depth / waves");
    __shared__ __attribute__((aligned(1024))) char lds_all[LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: everything derived from it stays scalar
    DCF_STAMP(0);
    char *ldsA = lds_all + wid * WAVE_LDS;
    char *ldsB = ldsA + NS * SA;
    const unsigned ldsA0 = lds_addr(ldsA), ldsB0 = lds_addr(ldsB);

    // XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs and this kernel runs one workgroup per
    // CU, so XCD x = blockIdx % 8 owns the x-th contiguous run of the unit list, a unit being the three kernel rows of
    // one (position range, co tile, ci tile), ranges slowest: the workgroups that re-read a range's gy / x rows
    // (3 * co_tiles * ci_tiles of them) sit on one or two XCDs and find them in that L2 instead of each of the eight
    // XCDs pulling its own copy of every range over the fabric.
    const int tiles2 = a.co_tiles * a.ci_tiles;
    // (round 5: units per XCD from the host -- dcf_wgrad_upx -- and a layer's first run goes to XCD a.xcd, as in conv_wgs.hip)
    const int units = tiles2 * a.nsplit, upx = a.upx;
    const int slot = bid >> 3;
    const int unit = (((bid & 7) - a.xcd) & 7) * upx + slot / 3;
    if (unit >= units) return;
    const int ki = slot % 3;
    const int slab_id = unit / tiles2;
    const int t2 = unit - slab_id * tiles2;
    const int cit = t2 % a.ci_tiles;
    const int cot = t2 / a.ci_tiles;
    const int co0 = cot * TM * 32, ci0 = cit * TN * 32;
    const int Wp = a.Wo + 2, BH = a.B * a.Ho;
    const int split = slab_id * NW + wid;
    const int q_begin = split * a.per_split;
    const int q_end = min(q_begin + a.per_split, a.M);
    const int nst = q_begin < q_end ? (q_end - q_begin + PK - 1) / PK : 0;

    f32x16 acc[3][TM][TN];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[k][i][j][q] = 0.f;
    const bool do_sum = (a.gsum != nullptr) && (ki == 0) && (cit == 0);
    float fsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) fsum[i] = 0.f;

    // ---- DMA side.  Lane = (row lr, 16-B chunk ch) of every instruction; the XOR swizzle goes on the source chunk.
    // The stage's first position is wave-uniform, so its (image row, padded column) live in SGPRs; a lane adds its
    // row and handles the single possible wrap into the next image row (Wo + 2 >= rows staged).
    const int rowA = a.Cout * 2;
    const int coutA = min(TM * 32, a.Cout - co0) * 2, cinB = min(TN * 32, a.Cin - ci0) * 2;
    const int lrA = lane / LPA, lrB = lane / LPB;
    const int chA = (RA == 128) ? ((lane % LPA) ^ (((lrA >> 1) & 1) << 2)) : (lane % LPA);
    const int chB = (RB == 128) ? ((lane % LPB) ^ (((lrB >> 1) & 1) << 2)) : (lane % LPB);
    const bool chokA = (chA * 16 < coutA) & !(DCF_DBG(a) & 1), chokB = (chB * 16 < cinB) & !(DCF_DBG(a) & 1);
    const __amdgpu_buffer_rsrc_t srcG = __builtin_amdgcn_make_buffer_rsrc((void *)a.gy, 0, a.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int laneA = opaque(co0 * 2 + chA * 16 + lrA * rowA);
    const int laneB = opaque(ci0 * 2 + chB * 16 + lrB * a.pixbytes);
    const int lrAo = opaque(lrA), lrBo = opaque(lrB);
    const int wrapA = opaque(2 * rowA), wrapB = opaque(2 * a.pixbytes);
    int sq = q_begin;                                    // scalar state: gy rows start at position sq, x rows at sq - 1
    int sR = q_begin / Wp, sC = q_begin - sR * Wp;
    int xR, xC, xOh;
    if (q_begin == 0) { xR = -1; xC = Wp - 1; xOh = a.Ho - 1; }
    else { xR = (q_begin - 1) / Wp; xC = (q_begin - 1) - xR * Wp; xOh = xR % a.Ho; }
    auto issue = [&](int is) {
        const unsigned sa = __builtin_amdgcn_readfirstlane(ldsA0 + is * SA), sb = __builtin_amdgcn_readfirstlane(ldsB0 + is * SB);
        {
            const int rem = q_end - sq;
            const int base = __builtin_amdgcn_readfirstlane((sR * a.Wo + sC - 1) * rowA);   // may be < 0 (lane rows make it valid); tensors < 2 GiB
            const int cl = sC + lrAo;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int c = cl + j * RPA;
                const bool w = c >= Wp;
                const int cc = w ? c - Wp : c;
                const bool ok = chokA & (lrAo < rem - j * RPA) & ((unsigned)(cc - 1) < (unsigned)a.Wo);
                const int off = laneA + (base + j * RPA * rowA) - (w ? wrapA : 0);
                glds16(srcG, ok ? (unsigned)off : OOB, sa + j * 1024);
            }
        }
        {
            const int ih0 = xOh + ki - 1;
            const int oh1 = xOh + 1 == a.Ho ? 0 : xOh + 1;
            const int ih1 = oh1 + ki - 1;
            const bool ok0 = (xR >= 0) & (xR < BH) & (ih0 >= 0) & (ih0 < a.H);
            const bool ok1 = (xR + 1 < BH) & (ih1 >= 0) & (ih1 < a.H);
            const int base = __builtin_amdgcn_readfirstlane(((xR + ki - 1) * a.W + xC - 1) * a.pixbytes);
            const int cl = xC + lrBo;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int c = cl + j * RPB;
                const bool w = c >= Wp;
                const int cc = w ? c - Wp : c;
                const bool ok = chokB & (lrBo < XROWS - j * RPB) & ((unsigned)(cc - 1) < (unsigned)a.Wo) & (w ? ok1 : ok0);
                const int off = laneB + (base + j * RPB * a.pixbytes) - (w ? wrapB : 0);
                glds16(srcX, ok ? (unsigned)off : OOB, sb + j * 1024);
            }
        }
        sq += PK;
        sC += PK;
        if (sC >= Wp) { sC -= Wp; ++sR; }
        xC += PK;
        if (xC >= Wp) { xC -= Wp; ++xR; xOh = xOh + 1 == a.Ho ? 0 : xOh + 1; }
    };

    // ---- read side: lane-constant byte offsets of the transposed reads (swizzle folded in)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g >> 1;
    const int colw = (16 * (g & 1) + 4 * pp) * 2;
    int offA[TM], offB[3][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = opaque((8 * hh + q) * RA + ((i * 64 + colw) ^ ((RA == 128) ? (((q >> 1) & 1) << 6) : 0)));
#pragma unroll
    for (int kj = 0; kj < 3; ++kj)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            offB[kj][j] = opaque((8 * hh + q + kj) * RB + ((j * 64 + colw) ^ ((RB == 128) ? ((((q + kj) >> 1) & 1) << 6) : 0)));

#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nst) issue(s0);
    int rslot = 0, islot = NS - 1;
    DCF_STAMP(1);
    for (int n = 0; n < nst; ++n) {
        if (n + NS - 1 < nst) issue(islot);
        const int ahead = min(NS - 1, nst - 1 - n);          // stages issued after stage n
        if (ahead >= NS - 1) wait_vmcnt<(NS - 1) * GI>();
        else if (NS > 3 && ahead == 2) wait_vmcnt<2 * GI>();
        else if (NS > 2 && ahead == 1) wait_vmcnt<1 * GI>();
        else wait_vmcnt<0>();
        const char *pa = ldsA + rslot * SA, *pb = ldsB + rslot * SB;
#pragma unroll
        for (int ks = 0; ks < PK / 16; ++ks) {
            uint4 fa[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const char *base = pa + offA[i] + ks * 16 * RA;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * RA));
                uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) {
                uint4 fb[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const char *base = pb + offB[kj][j] + ks * 16 * RB;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * RB));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        Mma<T>::run(fa[i], fb[j], acc[kj][i][j]);
            }
            if (do_sum) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const unsigned w[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float lo, hi; unpack2<T>(w[e], lo, hi); fsum[i] += lo + hi; }
                }
            }
        }
        rslot = rslot + 1 == NS ? 0 : rslot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }

    float *red = reinterpret_cast<float *>(lds_all);
    DCF_STAMP(2);
    __syncthreads();                                   // every wave is done with its ring
    DCF_STAMP(3);
    if (a.gsum != nullptr && ki == 0 && cit == 0) {    // block-uniform.  gsum rows: 4 per slab (NW = 8: waves w and w + 4 pair up)
        float tot[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) tot[i] = fsum[i] + __shfl_xor(fsum[i], 32, 64);
        if (NW == 8) {
            if (wid >= 4 && lane < 32)
#pragma unroll
                for (int i = 0; i < TM; ++i) red[((wid - 4) * TM + i) * 32 + lane] = tot[i];
            __syncthreads();
            if (wid < 4 && lane < 32)
#pragma unroll
                for (int i = 0; i < TM; ++i) tot[i] += red[(wid * TM + i) * 32 + lane];
            __syncthreads();
        }
        if (wid < 4 && lane < 32)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int co = co0 + i * 32 + lane;
                if (co < a.Cout) a.gsum[(size_t)(slab_id * 4 + wid) * a.Cout + co] = tot[i];
            }
    }
    // Cross-wave reduction, one tap at a time: every wave parks its tile in LDS (b128, lane-linear), then wave w sums
    // slice w of the NW copies in the fixed order w0 + w1 + ... and stores it -- all waves store in parallel, and the
    // barriers are raw (LDS-only wait): a __syncthreads() here would also drain the previous tap's global stores.
    float *slab = a.slabs + (size_t)slab_id * a.Cout * 9 * a.Cin;
    const int r = lane & 31, h = lane >> 5;
    constexpr int NG = TM * TN * 4;                   // float4 groups (4 consecutive accumulator registers) per tile
    static_assert(NG % NW == 0 || NW % NG == 0, "slices");
    float4 *red4 = reinterpret_cast<float4 *>(lds_all);
#pragma unroll
    for (int kj = 0; kj < 3; ++kj) {
        if (kj) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x16 &v = acc[kj][i][j];
                    red4[(wid * NG + (i * TN + j) * 4 + g4) * 64 + lane] = make_float4(v[4 * g4], v[4 * g4 + 1], v[4 * g4 + 2], v[4 * g4 + 3]);
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int tap = ki * 3 + kj;
        for (int grp = wid; grp < NG; grp += NW) {     // NG = 16, 8 or 4 groups over NW = 8 or 4 waves
            float4 sum = red4[grp * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const float4 v = red4[(w * NG + grp) * 64 + lane];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const int tile = grp >> 2, g4 = grp & 3;
            const int i = tile / TN, j = tile - i * TN;
            const int ci = ci0 + j * 32 + r;
            const int co = co0 + i * 32 + 8 * g4 + 4 * h;    // registers 4*g4 .. 4*g4+3 are rows co .. co+3
            if (ci < a.Cin) {
                float *dst = slab + ((size_t)co * 9 + tap) * a.Cin + ci;
                const size_t rs = (size_t)9 * a.Cin;
                if (co + 0 < a.Cout) dst[0] = sum.x;
                if (co + 1 < a.Cout) dst[rs] = sum.y;
                if (co + 2 < a.Cout) dst[2 * rs] = sum.z;
                if (co + 3 < a.Cout) dst[3 * rs] = sum.w;
            }
        }
    }
    DCF_STAMP(4);
}

template <typename T, int TM, int TN, int NS, int NW>
__global__ void __launch_bounds__(NW * 64) k_conv_wgrad3g(WgArgs a)
{
    wgrad3g_body<T, TM, TN, NS, NW>(a, blockIdx.x);
}

// The weight gradients of a backward pass do not depend on each other: up to DCF_WG_GROUP layers in ONE launch (the
// arguments travel in the kernel-argument segment).  Workgroups of the next layer start on CUs as the previous layer's
// finish, so there is no drain / launch bubble and no idle tail between layers.  Per-layer grids are multiples of 8
// workgroups, which keeps every layer's XCD mapping (workgroup index mod 8) intact.
#define DCF_WG_GROUP 32
struct WgGroup {
    WgArgs a[DCF_WG_GROUP];
    int off[DCF_WG_GROUP + 1];     // first workgroup of each layer
    int n;
};

template <typename T, int TM, int TN, int NS, int NW>
__global__ void __launch_bounds__(NW * 64) k_conv_wgrad3g_grp(WgGroup g)
{
    int i = 0;
#pragma unroll
    for (int k = 1; k < DCF_WG_GROUP; ++k) i += (k < g.n && (int)blockIdx.x >= g.off[k]);
    wgrad3g_body<T, TM, TN, NS, NW>(g.a[i], (int)blockIdx.x - g.off[i]);
}

// same grouping for the generic weight-gradient kernel (stride-2, 1x1, fusion GEMM layers)
template <typename T, int TM, int TN>
__global__ void __launch_bounds__(256) k_conv_wgrad_grp(WgGroup g)
{
    int i = 0;
#pragma unroll
    for (int k = 1; k < DCF_WG_GROUP; ++k) i += (k < g.n && (int)blockIdx.x >= g.off[k]);
    wgrad_body<T, TM, TN>(g.a[i], (int)blockIdx.x - g.off[i]);
}

}  // namespace

// row-sharing kernel for the 3x3 / stride-1 / pad-1 layers (conv_rs.hip)
int dcf_conv3x3_rs_launch(int dtype, const void *x, const void *w, const float *shift, const void *res, const void *mask, void *y,
                          int B, int H, int W, int Ck, int Cn, int relu, int flip, const char *name_base, double flops, hipStream_t s);
// the same layers with loader and consumer waves (conv_lc.hip); option CONV_LC=0 falls back to conv_rs.hip
int dcf_conv3x3_lc_launch(int dtype, const void *x, const void *w, const float *shift, const void *res, const void *mask, void *y,
                          int B, int H, int W, int Ck, int Cn, int relu, int flip, const char *name_base, double flops, int force, hipStream_t s);
// CONV_LC: 0 = never, 2 = whenever the shape is supported, unset / 1 = where it measured faster (see dcf_conv3x3_lc_launch)
static int use_lc()
{
    static DcfOpt e_o("CONV_LC"); const char *e = e_o.str();
    return e ? atoi(e) : 1;
}
// spatial-tile streaming kernel for the HBM-bound 32 / 64-channel 3x3 / stride-1 layers (conv_sp.hip)
int dcf_conv3x3_sp_launch(int dtype, const void *x, const void *w, const float *shift, const void *res, const void *mask, void *y,
                          int B, int H, int W, int Ck, int Cn, int relu, int flip, const char *name_base, double flops, hipStream_t s);
static bool use_rs(int dtype, int kh, int kw, int stride, int pad)
{
    static DcfOpt e_o("CONV_RS"); const char *e = e_o.str();
    return !(e && atoi(e) == 0) && dtype != DCF_F32 && kh == 3 && kw == 3 && stride == 1 && pad == 1;
}

// shared-staging weight gradient of the 3x3 / stride-1 layers with 128- or 192-channel tiles (conv_wgs.hip)
struct dcf_wgs_item {
    const void *x, *gy;
    float *slabs, *gsum;
    int B, H, W, Cin, Cout, nsplit;
};
int dcf_wgrad3v_splits(int B, int H, int W, int Cin, int Cout);
int dcf_wgrad3v_launch(int dtype, const void *x, const void *gy, float *slabs, float *gsum, int nsplit, int B, int H, int W, double flops, hipStream_t s);
int dcf_wgrad3s_kind(int dtype, int B, int H, int W, int Cin, int Cout);
int dcf_wgrad3s_splits(int kind, int B, int H, int W, int Cin, int Cout);
int dcf_wgrad3s_launch(int dtype, int kind, const dcf_wgs_item *items, int n, double flops, double bytes, hipStream_t s);
// shared-staging kernel for the 1x1 (any stride) and 3x3 / stride-2 layers (conv_wg1.hip)
struct dcf_wg1_item {
    const void *x, *gy;
    float *slabs, *gsum;
    int B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, nsplit;
};
int dcf_wgrad1s_kind(int dtype, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad);
int dcf_wgrad1s_splits(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw);
int dcf_wgrad1s_launch(int dtype, const dcf_wg1_item *items, int n, double flops, double bytes, hipStream_t s);

// ================================================================== C ABI
static int check_conv(const char *who, int dtype, int Cin, int Cout, int kh, int kw, int stride)
{
    const int es = dtype == DCF_F32 ? 4 : 2;
    DCF_REQUIRE(dtype == DCF_F32 || dtype == DCF_BF16 || dtype == DCF_F16, "%s: unsupported dtype %d", who, dtype);
    DCF_REQUIRE((Cin * es) % 64 == 0, "%s: Cin*esize must be a multiple of 64 bytes (Cin=%d)", who, Cin);
    DCF_REQUIRE(Cout % 32 == 0, "%s: Cout must be a multiple of 32 (Cout=%d)", who, Cout);
    DCF_REQUIRE(kh >= 1 && kw >= 1 && kh <= 7 && kw <= 7, "%s: kernel size %dx%d unsupported", who, kh, kw);
    DCF_REQUIRE(stride == 1 || stride == 2, "%s: stride %d unsupported", who, stride);
    return DCF_OK;
}

static int conv2d_fwd_impl(int dtype, const void *x, const void *w, const float *shift, const float *rowscale, const void *res, void *y,
                           int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                           int relu, dcf_stream_t stream);

extern "C" int dcf_conv2d_fwd(int dtype, const void *x, const void *w, const float *shift, const void *res, void *y,
                              int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                              int relu, dcf_stream_t stream)
{
    return conv2d_fwd_impl(dtype, x, w, shift, nullptr, res, y, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, relu, stream);
}

// The same with the shift scaled per output pixel: y = act(conv(x, w) + rowscale[m] * shift[c] + res), rowscale fp32 [B*Ho*Wo].
// The fusion site's second Linear layer under the neighbour sum (model.py:216-219): sum_k (W2 h_k + b2) = W2 sum_k h_k + cnt * b2
// -- the cnt * b2 term in the GEMM's epilogue instead of a pass of its own over the result.  1x1 layers only.
extern "C" int dcf_conv2d_fwd_rowscale(int dtype, const void *x, const void *w, const float *shift, const float *rowscale, const void *res,
                                       void *y, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                       int relu, dcf_stream_t stream)
{
    DCF_REQUIRE(shift && rowscale && kh == 1 && kw == 1, "dcf_conv2d_fwd_rowscale: needs shift, rowscale and a 1x1 layer");
    return conv2d_fwd_impl(dtype, x, w, shift, rowscale, res, y, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, relu, stream);
}

static int conv2d_fwd_impl(int dtype, const void *x, const void *w, const float *shift, const float *rowscale, const void *res, void *y,
                           int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                           int relu, dcf_stream_t stream)
{
    int rc = check_conv("dcf_conv2d_fwd", dtype, Cin, Cout, kh, kw, stride);
    if (rc) return rc;
    DCF_REQUIRE(x && w && y, "dcf_conv2d_fwd: null pointer");
    DCF_REQUIRE(Ho == (H + 2 * pad - kh) / stride + 1 && Wo == (W + 2 * pad - kw) / stride + 1, "dcf_conv2d_fwd: output size mismatch");
    DCF_REQUIRE((int64_t)B * H * W * Cin < (1ll << 31) * 1, "dcf_conv2d_fwd: tensor too large for 32-bit pixel index");
    ConvArgs a;
    a.x = (const char *)x; a.w = (const char *)w; a.shift = shift; a.res = (const char *)res; a.y = (char *)y;
    a.mask = nullptr; a.parity = 0; a.rowscale = rowscale;
    a.B = B; a.Hi = H; a.Wi = W; a.Ck = Cin; a.Ho = Ho; a.Wo = Wo; a.Cn = Cout;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad; a.relu = relu; a.M = B * Ho * Wo;
    a.pixbytes = Cin * (dtype == DCF_F32 ? 4 : 2);
    DCF_REQUIRE((int64_t)B * H * W * a.pixbytes < 0xFFFFFF00ll, "dcf_conv2d_fwd: tensor exceeds the 4 GiB buffer-descriptor range");
    a.xbytes = (unsigned)((int64_t)B * H * W * a.pixbytes);
    a.wbytes = (unsigned)((int64_t)Cout * kh * kw * a.pixbytes);
    const double flops = 2.0 * a.M * Cout * (double)Cin * kh * kw;
    if (use_rs(dtype, kh, kw, stride, pad)) {
        rc = dcf_conv3x3_sp_launch(dtype, x, w, shift, res, nullptr, y, B, H, W, Cin, Cout, relu, 0, dtype == DCF_F16 ? "conv_fwd_f16" : "conv_fwd_bf16", flops, S(stream));
        if (rc != DCF_EUNSUPPORTED) return rc;
        if (use_lc()) {
            rc = dcf_conv3x3_lc_launch(dtype, x, w, shift, res, nullptr, y, B, H, W, Cin, Cout, relu, 0, dtype == DCF_F16 ? "conv_fwd_f16" : "conv_fwd_bf16", flops, use_lc() == 2, S(stream));
            if (rc != DCF_EUNSUPPORTED) return rc;
        }
        rc = dcf_conv3x3_rs_launch(dtype, x, w, shift, res, nullptr, y, B, H, W, Cin, Cout, relu, 0, dtype == DCF_F16 ? "conv_fwd_f16" : "conv_fwd_bf16", flops, S(stream));
        if (rc != DCF_EUNSUPPORTED) return rc;
    }
    if (dtype == DCF_F32) return launch_igemm<float, false>(a, S(stream), "conv_fwd_f32", flops);
    if (dtype == DCF_F16) return launch_igemm<f16_t, false>(a, S(stream), "conv_fwd_f16", flops);
    return launch_igemm<bf16_t, false>(a, S(stream), "conv_fwd_bf16", flops);
}

static int conv2d_dgrad_impl(int dtype, const void *gy, const void *wt, const void *res, const void *resq, const void *mask, void *gx,
                             int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                             dcf_stream_t stream)
{
    // roles swap: reduction channels = Cout, produced channels = Cin
    int rc = check_conv("dcf_conv2d_dgrad", dtype, Cout, Cin, kh, kw, stride);
    if (rc) return rc;
    DCF_REQUIRE(gy && wt && gx, "dcf_conv2d_dgrad: null pointer");
    ConvArgs a;
    a.x = (const char *)gy; a.w = (const char *)wt; a.shift = nullptr; a.res = (const char *)res; a.y = (char *)gx;
    a.mask = (const char *)mask; a.resq = (const char *)resq;
    static DcfOpt par_env_o("DGRAD_PARITY"); const char *par_env = par_env_o.str();     // experiments: 0 = off, 1 = all stride-2 layers, 2 = full-line pixels only
    const int par_mode = par_env ? atoi(par_env) : 1;
    // (a 1x1 stride-2 layer has one live class and three that only store zeros: one plain pass is cheaper)
    a.parity = (stride == 2 && kh * kw > 1 && par_mode && (par_mode == 1 || Cin * (dtype == DCF_F32 ? 4 : 2) >= 128)) ? 1 : 0;
    for (int p = 0; p < 2; ++p) {       // rows / columns of dX whose (index + pad) has parity p
        a.cls_h0[p] = (p + pad) & 1; a.cls_w0[p] = (p + pad) & 1;
        a.cls_h[p] = (H - a.cls_h0[p] + 1) / 2; a.cls_w[p] = (W - a.cls_w0[p] + 1) / 2;
    }
    DCF_REQUIRE(!resq || a.parity, "dcf_conv2d_dgrad_halfres: needs a stride-2 layer with a kernel wider than 1x1 (and DGRAD_PARITY on)");
    a.B = B; a.Hi = Ho; a.Wi = Wo; a.Ck = Cout; a.Ho = H; a.Wo = W; a.Cn = Cin;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad; a.relu = 0; a.M = B * H * W;
    a.pixbytes = Cout * (dtype == DCF_F32 ? 4 : 2);
    DCF_REQUIRE((int64_t)B * Ho * Wo * a.pixbytes < 0xFFFFFF00ll, "dcf_conv2d_dgrad: tensor exceeds the 4 GiB buffer-descriptor range");
    a.xbytes = (unsigned)((int64_t)B * Ho * Wo * a.pixbytes);
    a.wbytes = (unsigned)((int64_t)Cin * kh * kw * a.pixbytes);
    const double flops = 2.0 * B * Ho * Wo * Cout * (double)Cin * kh * kw;   // algorithmic (= the forward conv's)
    if (use_rs(dtype, kh, kw, stride, pad)) {
        rc = dcf_conv3x3_sp_launch(dtype, gy, wt, nullptr, res, mask, gx, B, H, W, Cout, Cin, 0, 1, dtype == DCF_F16 ? "conv_dgrad_f16" : "conv_dgrad_bf16", flops, S(stream));
        if (rc != DCF_EUNSUPPORTED) return rc;
        if (use_lc()) {
            rc = dcf_conv3x3_lc_launch(dtype, gy, wt, nullptr, res, mask, gx, B, H, W, Cout, Cin, 0, 1, dtype == DCF_F16 ? "conv_dgrad_f16" : "conv_dgrad_bf16", flops, use_lc() == 2, S(stream));
            if (rc != DCF_EUNSUPPORTED) return rc;
        }
        rc = dcf_conv3x3_rs_launch(dtype, gy, wt, nullptr, res, mask, gx, B, H, W, Cout, Cin, 0, 1, dtype == DCF_F16 ? "conv_dgrad_f16" : "conv_dgrad_bf16", flops, S(stream));
        if (rc != DCF_EUNSUPPORTED) return rc;
    }
    if (dtype == DCF_F32) return launch_igemm<float, true>(a, S(stream), "conv_dgrad_f32", flops);
    if (dtype == DCF_F16) return launch_igemm<f16_t, true>(a, S(stream), "conv_dgrad_f16", flops);
    return launch_igemm<bf16_t, true>(a, S(stream), "conv_dgrad_bf16", flops);
}

extern "C" int dcf_conv2d_dgrad(int dtype, const void *gy, const void *wt, const void *res, const void *mask, void *gx,
                                int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                dcf_stream_t stream)
{
    return conv2d_dgrad_impl(dtype, gy, wt, res, nullptr, mask, gx, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, stream);
}

// Input gradient of a stride-2 layer (kernel wider than 1x1) that also takes a residual living on the (2i, 2j) sub-grid of
// gx: resq = [B][ceil(H/2)][ceil(W/2)][Cin], zero everywhere else by construction.  That is what a 1x1 / stride-2 shortcut
// (model.py:27-30 `down_conv`) sends back to the block input: computed there as a dense GEMM on its own grid instead of a
// full-resolution tensor that is three quarters zeros.  gx = (dgrad + res + scatter(resq)) * (mask > 0).
extern "C" int dcf_conv2d_dgrad_halfres(int dtype, const void *gy, const void *wt, const void *res, const void *resq, const void *mask,
                                        void *gx, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride,
                                        int pad, dcf_stream_t stream)
{
    DCF_REQUIRE(resq, "dcf_conv2d_dgrad_halfres: null resq");
    return conv2d_dgrad_impl(dtype, gy, wt, res, resq, mask, gx, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, stream);
}

static void wgrad_tiles(int Cin, int Cout, int &TM, int &TN) { TM = Cout >= 64 ? 2 : 1; TN = Cin >= 64 ? 2 : 1; }

// 3x3 / stride-1 layers: tile of the row-sharing kernel; KR = kernel rows per wave
static int wgrad3_nw()
{
    static DcfOpt e_o("WGRAD3_NW"); const char *e = e_o.str();
    return (e && atoi(e) == 4) ? 4 : 8;
}

static bool wgrad3_dma(int Wo, int TM, int TN)
{
    static DcfOpt e_o("WGRAD3_DMA"); const char *e = e_o.str();
    if (e && atoi(e) == 0) return false;
    return (Wo + 2 >= (TN == 2 ? 40 : 48)) && (TM * TN >= 2);   // a stage's rows wrap into the next image row at most once
}

static bool wgrad3_tiles(int Cin, int Cout, int kh, int kw, int stride, int &TM, int &TN, int &KR)
{
    static DcfOpt off_o("WGRAD3"); const char *off = off_o.str();
    if (off && atoi(off) == 0) return false;
    if (!(kh == 3 && kw == 3 && stride == 1)) return false;
    TM = Cout >= 64 ? 2 : 1; TN = Cin >= 64 ? 2 : 1;
    KR = (TM == 1 && TN == 1) ? 3 : 1;
    return true;
}

// XCD-aware work mapping of the generic weight-gradient kernel (wgrad_body): the (channel tile, tap) workgroups of a pixel range
// all run on ONE XCD and read the range's gy / x rows out of its L2.  Needs nsplit to be a multiple of 8 (ranges are dealt to the
// XCDs eight at a time).  Round 4 used it from 48 ranges on ("a loss for the few-range layers" -- measured on launches that did not
// fill the chip); round 5's per-layer PMC pass (tools/wgrad_traffic.py, profiles/r05o_wgrad_traffic_layers.txt) found the 3x3 /
// stride-2 layers below that bound fetching 4-6.6 x their tensors (nine taps x 2-12 channel-tile pairs spread over eight L2s):
// for nine-tap layers the bound is WGRAD_XCD_MIN9 (default 8).
static int wgrad_xcd_min(int taps)
{
    static DcfOpt a_o("WGRAD_XCD_MIN"), b_o("WGRAD_XCD_MIN9");
    const char *a = a_o.str(), *b = b_o.str();
    return taps >= 9 ? (b ? atoi(b) : 8) : (a ? atoi(a) : 48);
}
// Units (= (pixel range, channel-tile pair)) per XCD of the row-sharing / shared-staging kernels: the units dealt evenly over the
// eight XCDs.  Option WGRAD_RANGE_XCD=1 (round 5, measured, not the default): whole pixel ranges per XCD -- every layer's fetch then
// falls to 1.0-1.1x of its tensors (profiles/r05v_wgrad_traffic_layers.txt: the step's weight gradients 2.17 GB = 1.05x, against
// 2.54 GB = 1.20x), but a layer with two or four ranges then runs on two or four XCDs and the grouped launches take LONGER: cfg2
// 5.32-5.33 ms against 5.21-5.23 (same box), cfg4 11.22 against 11.11 -- these launches wait for their staging, not for HBM.
int dcf_wgrad_upx(int tiles2, int nsplit)
{
    static DcfOpt o("WGRAD_RANGE_XCD"); const char *e = o.str();
    if (e && atoi(e) == 1) return tiles2 * cdiv(nsplit, 8);
    return cdiv(tiles2 * nsplit, 8);
}
static int wgrad_xcd(int nsplit, int taps) { return ((nsplit & 7) == 0 && nsplit >= wgrad_xcd_min(taps)) ? 1 : 0; }

extern "C" int dcf_conv2d_wgrad_splits(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride)
{
    int TM, TN, KR;
    int tiles;
    if (kh == 3 && kw == 3 && stride == 1) {      // (the 16-bit kernel's choice also for fp32 launches of the shape: any count works there)
        const int nv = dcf_wgrad3v_splits(B, Ho, Wo, Cin, Cout);
        if (nv) return nv;
        const int kind = dcf_wgrad3s_kind(DCF_BF16, B, Ho, Wo, Cin, Cout);
        if (kind) return dcf_wgrad3s_splits(kind, B, Ho, Wo, Cin, Cout);
    }
    // (1x1 and 3x3 / stride-2 layers of the 16-bit types: conv_wg1.hip; the padding is the usual k / 2 there -- a launch with
    // another one falls back to the generic kernel, which takes any split count)
    if (!(kh == 3 && kw == 3 && stride == 1) &&
        dcf_wgrad1s_kind(DCF_BF16, B, Ho * stride + kh, Wo * stride + kw, Cin, Ho, Wo, Cout, kh, kw, stride, kh / 2))
        return dcf_wgrad1s_splits(B, Ho, Wo, Cin, Cout, kh, kw);
    static DcfOpt gb_o("WGRAD_BLOCKS"); const char *gb = gb_o.str();
    // Workgroups per layer.  The backward issues the weight gradients in grouped launches (dcf_conv2d_wgrad_group), where
    // the layers overlap each other: a layer does not have to fill the chip on its own, and fewer pixel ranges mean fewer
    // slabs to write / reduce and fewer in-kernel epilogues.  Swept on cfg2 (40-step runs): 240/1024 -> 264.8 frames/s,
    // 160/512 -> 266.4, 120/512 -> 270.5, 80/512 -> 270.5, 120/384 -> 270.0.
    int64_t want_blocks = gb ? atoi(gb) : 512;
    bool dma = false;
    if (wgrad3_tiles(Cin, Cout, kh, kw, stride, TM, TN, KR)) {
        tiles = cdiv(Cout, TM * 32) * cdiv(Cin, TN * 32) * (3 / KR);
        static DcfOpt wb_o("WGRAD3_BLOCKS"); const char *wb = wb_o.str();
        // the LDS-DMA kernel runs one workgroup per CU: one wave of workgroups (floor, not ceil)
        dma = wgrad3_dma(Wo, TM, TN);
        // LDS-DMA kernel: one workgroup per CU, 3 workgroups per unit; 120 workgroups = 5 units per XCD (see below)
        if (dma) { want_blocks = wb ? atoi(wb) : 120; if (tiles <= want_blocks) want_blocks -= tiles - 1; }
        else want_blocks = 512;
    } else {
        wgrad_tiles(Cin, Cout, TM, TN);
        tiles = cdiv(Cout, TM * 32) * cdiv(Cin, TN * 32) * kh * kw;
    }
    const int64_t M = (int64_t)B * Ho * Wo;
    // nsplit = number of SLABS = workgroups along the pixel axis; each has 4 waves (own pixel ranges)
    int64_t want = cdiv(want_blocks, tiles);
    int64_t maxs = (M + 511) / 512;                   // at least 4 stages of 32 pixels per wave
    if (want > maxs) want = maxs;
    // keep each layer's slab arena small: it is written once and re-read by dcf_wgrad_finalize
    const int64_t slab_bytes = (int64_t)cdiv(Cout, 32) * 32 * kh * kw * Cin * 4;
    static DcfOpt cap_env_o("SLAB_CAP_MB"); const char *cap_env = cap_env_o.str();
    const int64_t cap = ((int64_t)(cap_env ? atoi(cap_env) : 16) << 20) / slab_bytes;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    // multiples of 8 enable the generic kernel's XCD-aware work mapping (wgrad_xcd)
    const bool generic = !wgrad3_tiles(Cin, Cout, kh, kw, stride, TM, TN, KR);
    const int xmin = wgrad_xcd_min(kh * kw);
    const int64_t r8 = (want + 4) / 8 * 8;
    if (generic && r8 >= 8 && r8 >= xmin) want = r8;
    else if (want >= 44 && !dma) want = r8;
    return (int)want;
}

extern "C" int dcf_conv2d_wgrad(int dtype, const void *x, const void *gy, float *slabs, float *gsum, int nsplit,
                                int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                dcf_stream_t stream)
{
    int rc = check_conv("dcf_conv2d_wgrad", dtype, Cin, Cout, kh, kw, stride);
    if (rc) return rc;
    DCF_REQUIRE(x && gy && slabs && nsplit > 0, "dcf_conv2d_wgrad: bad arguments");
    WgArgs a;
    a.dbg = dcf_ablate_opt("WGRAD3_DBG");
    a.x = (const char *)x; a.gy = (const char *)gy; a.slabs = slabs; a.gsum = gsum;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad;
    a.M = B * Ho * Wo; a.nsplit = nsplit;
    a.xcd = wgrad_xcd(nsplit, kh * kw);
    a.pixbytes = Cin * (dtype == DCF_F32 ? 4 : 2);
    DCF_REQUIRE((int64_t)B * H * W * a.pixbytes < 0xFFFFFF00ll && (int64_t)a.M * Cout * 4 < 0xFFFFFF00ll, "dcf_conv2d_wgrad: tensor exceeds the 4 GiB buffer-descriptor range");
    a.xbytes = (unsigned)((int64_t)B * H * W * a.pixbytes);
    a.gbytes = (unsigned)((int64_t)a.M * Cout * (dtype == DCF_F32 ? 4 : 2));
    hipStream_t s = S(stream);
    const double flops = 2.0 * a.M * Cout * (double)Cin * kh * kw;
    const double wbytes_ = (double)a.xbytes + (double)a.gbytes + (double)nsplit * Cout * kh * kw * Cin * 4.0;
    int TM, TN, KR;
    if (pad == 1 && H == Ho && W == Wo && kh == 3 && kw == 3 && stride == 1) {
        if (dtype != DCF_F32 && dcf_wgrad3v_splits(B, H, W, Cin, Cout)) return dcf_wgrad3v_launch(dtype, x, gy, slabs, gsum, nsplit, B, H, W, flops, s);
        const int kind = dcf_wgrad3s_kind(dtype, B, H, W, Cin, Cout);
        if (kind) {
            const dcf_wgs_item it = {x, gy, slabs, gsum, B, H, W, Cin, Cout, nsplit};
            const double wb = (double)a.xbytes + (double)a.gbytes + (double)nsplit * Cout * 9 * Cin * 4.0;
            return dcf_wgrad3s_launch(dtype, kind, &it, 1, flops, wb, s);
        }
    }
    if (dcf_wgrad1s_kind(dtype, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad)) {
        const dcf_wg1_item it = {x, gy, slabs, gsum, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, nsplit};
        return dcf_wgrad1s_launch(dtype, &it, 1, flops, wbytes_, s);
    }
    if (pad == 1 && H == Ho && W == Wo && wgrad3_tiles(Cin, Cout, kh, kw, stride, TM, TN, KR)) {
        const bool dma = dtype != DCF_F32 && wgrad3_dma(Wo, TM, TN) && (int64_t)B * H * W * a.pixbytes < (1ll << 31) && (int64_t)a.M * Cout * 2 < (1ll << 31);
        if (dtype == DCF_F32 && TM == 2 && TN == 2) TN = 1;   // fp32 accumulators + fragments of 2x2x3 do not fit 256 VGPRs
        a.M = B * Ho * (Wo + 2);                           // padded positions (see k_conv_wgrad3)
        a.per_split = cdiv(cdiv(a.M, 4 * nsplit), 32) * 32;
        a.co_tiles = cdiv(Cout, TM * 32);
        a.ci_tiles = cdiv(Cin, TN * 32);
        dim3 grid3(a.co_tiles * a.ci_tiles * (3 / KR) * nsplit);
        if (dma) {
            const int NW = wgrad3_nw();
            a.per_split = cdiv(cdiv(a.M, NW * nsplit), 32) * 32;
            a.upx = dcf_wgrad_upx(a.co_tiles * a.ci_tiles, nsplit);
            grid3 = dim3(8 * 3 * a.upx);    // see the kernel's work mapping
            a.xcd = 0;
#define DCF_WG3G_T(T_, N_, TM_, TN_)                                                                                                                    \
    do {                                                                                                                                                \
        if (NW == 8) DCF_LAUNCH_W("conv_wgrad3g_" N_ "<" #TM_ "," #TN_ ",2,8>", flops, s, hipLaunchKernelGGL((k_conv_wgrad3g<T_, TM_, TN_, 2, 8>), grid3, dim3(512), 0, s, a)); \
        else DCF_LAUNCH_W("conv_wgrad3g_" N_ "<" #TM_ "," #TN_ ",4,4>", flops, s, hipLaunchKernelGGL((k_conv_wgrad3g<T_, TM_, TN_, 4, 4>), grid3, dim3(256), 0, s, a));        \
    } while (0)
#define DCF_WG3G(TM_, TN_) do { if (dtype == DCF_F16) DCF_WG3G_T(f16_t, "f16", TM_, TN_); else DCF_WG3G_T(bf16_t, "bf16", TM_, TN_); } while (0)
            if (TM == 2 && TN == 2) DCF_WG3G(2, 2);
            else if (TM == 2) DCF_WG3G(2, 1);
            else DCF_WG3G(1, 2);
#undef DCF_WG3G
#undef DCF_WG3G_T
            if (DCF_DBG(a) & 2) {
                long long tt[8];
                (void)hipStreamSynchronize(s);
                (void)hipMemcpyFromSymbol(tt, HIP_SYMBOL(g_dcf_dbg_t), sizeof(tt));
                fprintf(stderr, "[wgrad3g %dx%d %d->%d ns=%d] clocks: prologue %lld loop %lld wait %lld epilogue %lld (100 MHz ticks x?)\n", Ho, Wo, Cin, Cout, nsplit,
                        tt[1] - tt[0], tt[2] - tt[1], tt[3] - tt[2], tt[4] - tt[3]);
            }
            return DCF_OK;
        }
#define DCF_WG3(T_, NAME_)                                                                                                                           \
    do {                                                                                                                                             \
        if (KR == 3) DCF_LAUNCH_WB(NAME_ "<1,1,3>", flops, wbytes_, s, hipLaunchKernelGGL((k_conv_wgrad3<T_, 1, 1, 3>), grid3, dim3(256), 0, s, a));            \
        else if (TM == 2 && TN == 2) DCF_LAUNCH_WB(NAME_ "<2,2,1>", flops, wbytes_, s, hipLaunchKernelGGL((k_conv_wgrad3<typename std::conditional<std::is_same<T_, float>::value, bf16_t, T_>::type, 2, 2, 1>), grid3, dim3(256), 0, s, a)); \
        else if (TM == 2) DCF_LAUNCH_WB(NAME_ "<2,1,1>", flops, wbytes_, s, hipLaunchKernelGGL((k_conv_wgrad3<T_, 2, 1, 1>), grid3, dim3(256), 0, s, a));       \
        else DCF_LAUNCH_WB(NAME_ "<1,2,1>", flops, wbytes_, s, hipLaunchKernelGGL((k_conv_wgrad3<T_, 1, 2, 1>), grid3, dim3(256), 0, s, a));                    \
    } while (0)
        if (dtype == DCF_F32) DCF_WG3(float, "conv_wgrad3_f32");
        else if (dtype == DCF_F16) DCF_WG3(f16_t, "conv_wgrad3_f16");
        else DCF_WG3(bf16_t, "conv_wgrad3_bf16");
#undef DCF_WG3
        return DCF_OK;
    }
    a.per_split = cdiv(cdiv(a.M, 4 * nsplit), 32) * 32;   // pixels per WAVE (4 waves reduce into one slab)
    wgrad_tiles(Cin, Cout, TM, TN);
    a.co_tiles = cdiv(Cout, TM * 32);
    a.ci_tiles = cdiv(Cin, TN * 32);
    dim3 grid(a.co_tiles * a.ci_tiles * kh * kw * nsplit);
#define DCF_WG(T_, NAME_)                                                                                                                       \
    do {                                                                                                                                        \
        if (TM == 2 && TN == 2) DCF_LAUNCH_WB(NAME_ "<2,2>", flops, wbytes_, s, hipLaunchKernelGGL((k_conv_wgrad<T_, 2, 2>), grid, dim3(256), 0, s, a));   \
        else if (TM == 2) DCF_LAUNCH_WB(NAME_ "<2,1>", flops, wbytes_, s, hipLaunchKernelGGL((k_conv_wgrad<T_, 2, 1>), grid, dim3(256), 0, s, a));         \
        else if (TN == 2) DCF_LAUNCH_WB(NAME_ "<1,2>", flops, wbytes_, s, hipLaunchKernelGGL((k_conv_wgrad<T_, 1, 2>), grid, dim3(256), 0, s, a));         \
        else DCF_LAUNCH_WB(NAME_ "<1,1>", flops, wbytes_, s, hipLaunchKernelGGL((k_conv_wgrad<T_, 1, 1>), grid, dim3(256), 0, s, a));                      \
    } while (0)
    if (dtype == DCF_F32) DCF_WG(float, "conv_wgrad_f32");
    else if (dtype == DCF_F16) DCF_WG(f16_t, "conv_wgrad_f16");
    else DCF_WG(bf16_t, "conv_wgrad_bf16");
#undef DCF_WG
    return DCF_OK;
}


// ---- grouped weight gradients (3x3 / stride 1 / pad 1 layers of the LDS-DMA kernel, 64x64 tiles)
extern "C" int dcf_conv2d_wgrad_groupable(int dtype, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad)
{
    static DcfOpt e_o("WGRAD_GROUP"); const char *e = e_o.str();
    if (e && atoi(e) == 0) return 0;
    int TM, TN, KR;
    if (dtype == DCF_F32 || pad != 1 || !wgrad3_tiles(Cin, Cout, kh, kw, stride, TM, TN, KR)) return 0;
    if (!(TM == 2 && TN == 2 && KR == 1) || !wgrad3_dma(W, TM, TN) || wgrad3_nw() != 8) return 0;
    const int64_t es = 2;
    return ((int64_t)B * H * W * Cin * es < (1ll << 31) && (int64_t)B * H * W * Cout * es < (1ll << 31)) ? 1 : 0;
}

extern "C" int dcf_conv2d_wgrad_group(const dcf_wgrad_item *items, int n, dcf_stream_t stream)
{
    DCF_REQUIRE(items && n >= 0, "dcf_conv2d_wgrad_group: bad arguments");
    hipStream_t s = S(stream);
    // bucket 0: row-sharing LDS-DMA kernel <2,2,2,8>; buckets 1..4: generic kernel <2,2> <2,1> <1,2> <1,1>; -1: on its own
    std::vector<int> bucket(n);
    for (int i = 0; i < n; ++i) {
        const dcf_wgrad_item &it = items[i];
        DCF_REQUIRE(it.x && it.gy && it.slabs && it.nsplit > 0, "dcf_conv2d_wgrad_group: item %d: bad arguments", i);
        const int Ho = (it.H + 2 * it.pad - it.kh) / it.stride + 1, Wo = (it.W + 2 * it.pad - it.kw) / it.stride + 1;
        int TM, TN, KR;
        if (it.dtype == DCF_F32) bucket[i] = -1;
        else if (it.pad == 1 && it.kh == 3 && it.kw == 3 && it.stride == 1 && dcf_wgrad3s_kind(it.dtype, it.B, it.H, it.W, it.Cin, it.Cout))
            bucket[i] = 4 + dcf_wgrad3s_kind(it.dtype, it.B, it.H, it.W, it.Cin, it.Cout);       // 5: shared-staging kernel
        else if (dcf_wgrad1s_kind(it.dtype, it.B, it.H, it.W, it.Cin, Ho, Wo, it.Cout, it.kh, it.kw, it.stride, it.pad)) bucket[i] = 6;   // conv_wg1.hip
        else if (dcf_conv2d_wgrad_groupable(it.dtype, it.B, it.H, it.W, it.Cin, it.Cout, it.kh, it.kw, it.stride, it.pad)) bucket[i] = 0;
        else if (it.pad == 1 && Ho == it.H && Wo == it.W && wgrad3_tiles(it.Cin, it.Cout, it.kh, it.kw, it.stride, TM, TN, KR)) bucket[i] = -1;
        else {
            wgrad_tiles(it.Cin, it.Cout, TM, TN);
            bucket[i] = 1 + (TM == 2 ? 0 : 2) + (TN == 2 ? 0 : 1);
        }
        if (bucket[i] < 0) {
            int rc = dcf_conv2d_wgrad(it.dtype, it.x, it.gy, it.slabs, it.gsum, it.nsplit, it.B, it.H, it.W, it.Cin, Ho, Wo, it.Cout, it.kh, it.kw,
                                      it.stride, it.pad, stream);
            if (rc) return rc;
        }
    }
    for (int bk = 5; bk <= 5; ++bk)
        for (int dt = DCF_BF16; dt <= DCF_F16; ++dt) {
            std::vector<dcf_wgs_item> ws;
            double flops = 0.0, bytes = 0.0;
            for (int i = 0; i < n; ++i) {
                const dcf_wgrad_item &it = items[i];
                if (bucket[i] != bk || it.dtype != dt) continue;
                ws.push_back({it.x, it.gy, it.slabs, it.gsum, it.B, it.H, it.W, it.Cin, it.Cout, it.nsplit});
                flops += 2.0 * it.B * it.H * it.W * (double)it.Cout * it.Cin * 9;
                bytes += (double)it.B * it.H * it.W * (it.Cin + it.Cout) * 2.0 + (double)it.nsplit * it.Cout * 9 * it.Cin * 4.0;
            }
            if (!ws.empty()) {
                int rc = dcf_wgrad3s_launch(dt, bk - 4, ws.data(), (int)ws.size(), flops, bytes, s);
                if (rc) return rc;
            }
        }
    for (int dt = DCF_BF16; dt <= DCF_F16; ++dt) {
        std::vector<dcf_wg1_item> ws;
        double flops = 0.0, bytes = 0.0;
        for (int i = 0; i < n; ++i) {
            const dcf_wgrad_item &it = items[i];
            if (bucket[i] != 6 || it.dtype != dt) continue;
            const int Ho = (it.H + 2 * it.pad - it.kh) / it.stride + 1, Wo = (it.W + 2 * it.pad - it.kw) / it.stride + 1;
            ws.push_back({it.x, it.gy, it.slabs, it.gsum, it.B, it.H, it.W, it.Cin, Ho, Wo, it.Cout, it.kh, it.kw, it.stride, it.pad, it.nsplit});
            flops += 2.0 * it.B * Ho * Wo * (double)it.Cout * it.Cin * it.kh * it.kw;
            bytes += ((double)it.B * it.H * it.W * it.Cin + (double)it.B * Ho * Wo * it.Cout) * 2.0 + (double)it.nsplit * it.Cout * it.kh * it.kw * it.Cin * 4.0;
        }
        if (!ws.empty()) {
            int rc = dcf_wgrad1s_launch(dt, ws.data(), (int)ws.size(), flops, bytes, s);
            if (rc) return rc;
        }
    }
    static DcfOpt gen_env_o("WGRAD_GROUP_GENERIC"); const char *gen_env = gen_env_o.str();
    const bool group_generic = !(gen_env && atoi(gen_env) == 0);
    for (int bk = 0; bk <= 4; ++bk) {
        for (int dt = DCF_BF16; dt <= DCF_F16; ++dt) {
            WgGroup g;
            int cnt = 0, blocks = 0, rot3 = 0;
            double flops = 0.0, bytes = 0.0;     // algorithmic: x and gy read once, one fp32 slab set written
            auto flush = [&]() -> int {
                if (cnt == 0) return DCF_OK;
                for (int k = cnt; k <= DCF_WG_GROUP; ++k) g.off[k] = blocks;
                for (int k = cnt; k < DCF_WG_GROUP; ++k) g.a[k] = g.a[0];
                g.n = cnt;
#define DCF_GRP_GEN(TM_, TN_)                                                                                                                      \
    do {                                                                                                                                           \
        if (dt == DCF_F16) DCF_LAUNCH_WB("conv_wgrad_grp_f16<" #TM_ "," #TN_ ">", flops, bytes, s, hipLaunchKernelGGL((k_conv_wgrad_grp<f16_t, TM_, TN_>), dim3(blocks), dim3(256), 0, s, g)); \
        else DCF_LAUNCH_WB("conv_wgrad_grp_bf16<" #TM_ "," #TN_ ">", flops, bytes, s, hipLaunchKernelGGL((k_conv_wgrad_grp<bf16_t, TM_, TN_>), dim3(blocks), dim3(256), 0, s, g)); \
    } while (0)
                if (bk == 0) {
                    if (dt == DCF_F16) DCF_LAUNCH_WB("conv_wgrad3g_grp_f16<2,2,2,8>", flops, bytes, s, hipLaunchKernelGGL((k_conv_wgrad3g_grp<f16_t, 2, 2, 2, 8>), dim3(blocks), dim3(512), 0, s, g));
                    else DCF_LAUNCH_WB("conv_wgrad3g_grp_bf16<2,2,2,8>", flops, bytes, s, hipLaunchKernelGGL((k_conv_wgrad3g_grp<bf16_t, 2, 2, 2, 8>), dim3(blocks), dim3(512), 0, s, g));
                } else if (bk == 1) DCF_GRP_GEN(2, 2);
                else if (bk == 2) DCF_GRP_GEN(2, 1);
                else if (bk == 3) DCF_GRP_GEN(1, 2);
                else DCF_GRP_GEN(1, 1);
#undef DCF_GRP_GEN
                cnt = 0; blocks = 0; flops = 0.0; bytes = 0.0;
                return DCF_OK;
            };
            for (int i = 0; i < n; ++i) {
                const dcf_wgrad_item &it = items[i];
                if (bucket[i] != bk || it.dtype != dt) continue;
                const int Ho = (it.H + 2 * it.pad - it.kh) / it.stride + 1, Wo = (it.W + 2 * it.pad - it.kw) / it.stride + 1;
                if (bk > 0 && !group_generic) {
                    int rc = dcf_conv2d_wgrad(it.dtype, it.x, it.gy, it.slabs, it.gsum, it.nsplit, it.B, it.H, it.W, it.Cin, Ho, Wo, it.Cout, it.kh,
                                              it.kw, it.stride, it.pad, stream);
                    if (rc) return rc;
                    continue;
                }
                WgArgs &a = g.a[cnt];
                a.dbg = 0;
                a.x = (const char *)it.x; a.gy = (const char *)it.gy; a.slabs = it.slabs; a.gsum = it.gsum;
                a.B = it.B; a.H = it.H; a.W = it.W; a.Cin = it.Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = it.Cout;
                a.kh = it.kh; a.kw = it.kw; a.stride = it.stride; a.pad = it.pad;
                a.nsplit = it.nsplit;
                a.xcd = wgrad_xcd(it.nsplit, it.kh * it.kw);
                a.pixbytes = it.Cin * 2;
                DCF_REQUIRE((int64_t)it.B * it.H * it.W * a.pixbytes < 0xFFFFFF00ll && (int64_t)it.B * Ho * Wo * it.Cout * 4 < 0xFFFFFF00ll,
                            "dcf_conv2d_wgrad_group: item %d: tensor exceeds the 4 GiB buffer-descriptor range", i);
                a.xbytes = (unsigned)((int64_t)it.B * it.H * it.W * a.pixbytes);
                a.gbytes = (unsigned)((int64_t)it.B * Ho * Wo * it.Cout * 2);
                g.off[cnt] = blocks;
                if (bk == 0) {
                    a.M = it.B * it.H * (it.W + 2);
                    a.per_split = cdiv(cdiv(a.M, 8 * it.nsplit), 32) * 32;
                    a.co_tiles = cdiv(it.Cout, 64);
                    a.ci_tiles = cdiv(it.Cin, 64);
                    const int upx3 = a.upx = dcf_wgrad_upx(a.co_tiles * a.ci_tiles, it.nsplit);
                    blocks += 8 * 3 * upx3;
                    a.xcd = rot3;
                    rot3 = (rot3 + cdiv(a.co_tiles * a.ci_tiles * it.nsplit, upx3)) & 7;
                } else {
                    int TM, TN;
                    wgrad_tiles(it.Cin, it.Cout, TM, TN);
                    a.M = it.B * Ho * Wo;
                    a.per_split = cdiv(cdiv(a.M, 4 * it.nsplit), 32) * 32;
                    a.co_tiles = cdiv(it.Cout, TM * 32);
                    a.ci_tiles = cdiv(it.Cin, TN * 32);
                    // the kernel's XCD mapping wants a layer to start on a multiple of 8 workgroups; the extra ones exit at once
                    blocks += (a.co_tiles * a.ci_tiles * it.kh * it.kw * it.nsplit + 7) / 8 * 8;
                }
                flops += 2.0 * it.B * Ho * Wo * (double)it.Cout * it.Cin * it.kh * it.kw;
                bytes += (double)a.xbytes + (double)a.gbytes + (double)it.nsplit * it.Cout * it.kh * it.kw * it.Cin * 4.0;
                if (++cnt == DCF_WG_GROUP) { int rc = flush(); if (rc) return rc; }
            }
            int rc = flush();
            if (rc) return rc;
        }
    }
    return DCF_OK;
}

// ------------------------------------------------------------------ image stem
// 7x7 stride-2 pad-3 convolution of the RGB image (SURVEY.md App. D image stream; the
// reference only has the commented-out torchvision resnet18 at model.py:192).
// The image is stored NHWC4 with a zero halo (dcf_image_to_nhwc4): one kernel row of
// 7 taps x 3 channels is then 8 pixels x 4 channels = 32 CONTIGUOUS elements, so the stem
// is the same implicit GEMM with 7 vertical taps of K=32, pixel pitch 4 elements, stride 2
// and no bounds checks.  Weights [Cout][7][8][4] (tap kw=7 and channel 3 are zero).
extern "C" int dcf_stem7x7_fwd(int dtype, const void *img4, const void *w, const float *shift, void *y,
                               int B, int H, int W, int Ho, int Wo, int Cout, int relu, dcf_stream_t stream)
{
    DCF_REQUIRE(img4 && w && y && Cout % 32 == 0, "dcf_stem7x7_fwd: bad arguments");
    DCF_REQUIRE(Ho == (H + 6 - 7) / 2 + 1 && Wo == (W + 6 - 7) / 2 + 1, "dcf_stem7x7_fwd: output size mismatch");
    ConvArgs a;
    a.x = (const char *)img4; a.w = (const char *)w; a.shift = shift; a.res = nullptr; a.y = (char *)y;
    a.mask = nullptr; a.parity = 0;
    a.B = B; a.Hi = H + 6; a.Wi = W + 8; a.Ck = 32; a.Ho = Ho; a.Wo = Wo; a.Cn = Cout;
    a.kh = 7; a.kw = 1; a.stride = 2; a.pad = 0; a.relu = relu; a.M = B * Ho * Wo;
    a.pixbytes = 4 * (dtype == DCF_F32 ? 4 : 2);
    a.xbytes = (unsigned)((int64_t)B * (H + 6) * (W + 8) * a.pixbytes);
    a.wbytes = (unsigned)((int64_t)Cout * 7 * 32 * (dtype == DCF_F32 ? 4 : 2));
    const double flops = 2.0 * a.M * Cout * 147.0;   // 7x7x3 taps (the padded K of 224 is not algorithmic work)
    if (dtype == DCF_F32) return launch_igemm<float, false>(a, S(stream), "stem_fwd_f32", flops);
    if (dtype == DCF_F16) return launch_igemm<f16_t, false>(a, S(stream), "stem_fwd_f16", flops);
    return launch_igemm<bf16_t, false>(a, S(stream), "stem_fwd_bf16", flops);
}

extern "C" int dcf_stem7x7_wgrad(int dtype, const void *img4, const void *gy, float *slabs, float *gsum, int nsplit,
                                 int B, int H, int W, int Ho, int Wo, int Cout, dcf_stream_t stream)
{
    DCF_REQUIRE(img4 && gy && slabs && nsplit > 0 && Cout % 32 == 0, "dcf_stem7x7_wgrad: bad arguments");
    WgArgs a;
    a.dbg = 0;
    a.x = (const char *)img4; a.gy = (const char *)gy; a.slabs = slabs; a.gsum = gsum;
    a.B = B; a.H = H + 6; a.W = W + 8; a.Cin = 32; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.kh = 7; a.kw = 1; a.stride = 2; a.pad = 0;
    a.M = B * Ho * Wo; a.nsplit = nsplit;
    a.xcd = wgrad_xcd(nsplit, 1);
    a.pixbytes = 4 * (dtype == DCF_F32 ? 4 : 2);
    a.xbytes = (unsigned)((int64_t)B * (H + 6) * (W + 8) * a.pixbytes);
    a.gbytes = (unsigned)((int64_t)a.M * Cout * (dtype == DCF_F32 ? 4 : 2));
    a.per_split = cdiv(cdiv(a.M, 4 * nsplit), 32) * 32;   // pixels per WAVE (4 waves reduce into one slab)
    const int TM = Cout >= 64 ? 2 : 1;
    a.co_tiles = cdiv(Cout, TM * 32);
    a.ci_tiles = 1;
    dim3 grid(a.co_tiles * 7 * nsplit);
    hipStream_t s = S(stream);
    const double sflops = 2.0 * a.M * Cout * 147.0;
    if (dtype == DCF_F32) {
        if (TM == 2) DCF_LAUNCH_W("stem_wgrad_f32", sflops, s, hipLaunchKernelGGL((k_conv_wgrad<float, 2, 1>), grid, dim3(256), 0, s, a));
        else DCF_LAUNCH_W("stem_wgrad_f32", sflops, s, hipLaunchKernelGGL((k_conv_wgrad<float, 1, 1>), grid, dim3(256), 0, s, a));
    } else if (dtype == DCF_F16) {
        if (TM == 2) DCF_LAUNCH_W("stem_wgrad_f16", sflops, s, hipLaunchKernelGGL((k_conv_wgrad<f16_t, 2, 1>), grid, dim3(256), 0, s, a));
        else DCF_LAUNCH_W("stem_wgrad_f16", sflops, s, hipLaunchKernelGGL((k_conv_wgrad<f16_t, 1, 1>), grid, dim3(256), 0, s, a));
    } else {
        if (TM == 2) DCF_LAUNCH_W("stem_wgrad_bf16", sflops, s, hipLaunchKernelGGL((k_conv_wgrad<bf16_t, 2, 1>), grid, dim3(256), 0, s, a));
        else DCF_LAUNCH_W("stem_wgrad_bf16", sflops, s, hipLaunchKernelGGL((k_conv_wgrad<bf16_t, 1, 1>), grid, dim3(256), 0, s, a));
    }
    return DCF_OK;
}
