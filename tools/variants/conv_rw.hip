// conv_rw.hip -- row-sharing implicit GEMM with the WEIGHTS IN REGISTERS, for the 3x3 / stride-1 / pad-1 convolutions
// (forward and input gradient), 16-bit element types, gfx950.
//
// The layers are those of conv_rs.hip (/root/reference/model.py:15-28 ResidualBlock bodies, :153 conv3; the camera trunk's
// BasicBlocks).  conv_rs.hip pushes BOTH operands through LDS-DMA rings and synchronises once per tap; what a CU can take in
// LDS-DMA pieces (60-180 cycles of issue each) and the wait / barrier / issue sequence around every tap held it at 0.26-0.31
// of the MFMA rate (profiles/r04a_sq_*.csv).  Here:
//
//   * WEIGHTS NEVER TOUCH LDS.  A weight element is used by ONE wave of the workgroup (the wave owns 32 output channels),
//     so LDS buys it nothing.  dcf_conv3x3_weight_frag() rewrites the [Cn][9][Ck] image in MFMA A-fragment order:
//         block (channel tile ct, tap, 64-channel chunk cc, k-step ks) = 1 KiB = 64 lanes x 16 B,
//         lane (r, h) = channel 32 ct + r, elements k = 64 cc + 16 ks + 8 h + {0..7},
//     and a wave loads a tap's fragments with four fully coalesced buffer_load_dwordx4 (L2 -> VGPR), PF taps ahead of their
//     use, into a rotating set of registers -- no LDS write, no LDS read, no LDS-DMA issue for half of the staged bytes.
//   * PIXELS keep the row-sharing LDS-DMA ring (one staged tile of BM + 2 padded positions per kernel row x 64-channel chunk,
//     read at row offsets 0 / 1 / 2 by the three horizontal taps), now DX stages ahead in DX + 1 slots: the LDS the weight
//     ring took is free.
//   * ONE BARRIER PER STAGE (three taps), and not where the reads wait for it: after tap 0 of stage s every wave waits for
//     its pieces of stage s + 1 and joins the barrier, so the fragment reads run one k-step ahead of the MFMAs across tap AND
//     stage boundaries; the barrier also tells that everyone has left stage s - 1, whose slot the DMA of stage s + DX
//     (issued in taps 1 and 2) overwrites.
//   * Everything a wave issues per k-step is a compile-time constant (surplus loads / pieces go out of range: zeros), so every
//     s_waitcnt vmcnt(N) is an immediate (rw_waits).
//
// Same K order as conv_rs.hip (kernel row, chunk, tap, k-step) and the same fragments: the results are bit-identical to it.
// dgrad = the same kernel on the fragment image of [Cin][tap][Cout] with the taps mirrored.
// Algorithmic work per launch: 2*B*H*W*Cout*Cin*9 flop; bytes B*H*W*(Cin + Cout)*2 + weights (+ residual / mask reads).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "dcf_common.h"          // -I <pkg>/csrc (tools/rw_variants.sh)
#include "conv_common.h"
#include "conv_rw.h"             // the four entry points: not part of include/dcf_hip.h

// Timing ablations, COMPILE-TIME only (tools/rw_variants.sh builds one library per mask; a run-time switch changed the register
// allocation of the whole kernel -- the 128-channel kind spilled -- and measured a different kernel): 1 no MFMAs, 2 pixel DMA
// reads nothing, 4 no epilogue, 16 weight loads read nothing, 32 pixel DMA not issued, 64 weight loads not issued, 128 no LDS
// fragment reads.  Results are wrong in those builds; the shipped library is built with 0.
#ifndef RW_DBG
#define RW_DBG 0
#endif

namespace {

struct RwArgs {
    const char *x;        // [B][H][W][Ck]
    const char *wf;       // fragment-ordered weights (dcf_conv3x3_weight_frag)
    const float *shift;   // [Cn] or null
    const char *res;      // [B*H*W][Cn] or null
    const char *mask;     // [B*H*W][Cn] or null: output *= (mask > 0)
    char *y;              // [B*H*W][Cn]
    int B, H, W, Ck, Cn;
    int relu, flip;       // flip = 1: input gradient (taps mirrored)
    int npt;              // 32-position tiles per workgroup
    int mtiles;           // position tiles of the launch
    int Q;                // padded positions B*H*(W+2)
    unsigned xbytes, wbytes, ybytes;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// One weight fragment: 64 lanes x 16 B, global / L2 -> VGPR.  Inline asm with our own counted waits: a compiler-visible load
// would make hipcc wait for it with a vmcnt that also drains the LDS-DMA pieces issued before it.
template <int IMM> __device__ __forceinline__ void gldw(u32x4 &d, __amdgpu_buffer_rsrc_t rsrc, unsigned voff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(d) : "v"(voff), "s"(rsrc), "n"(IMM));
}
__device__ __forceinline__ void rw_bind(u32x4 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void rw_keep(const f32x16 &v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void rw_gst16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, uint4 data)
{
    const u32x4 d = {data.x, data.y, data.z, data.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)voff, 0, 0);
}
template <typename T> __device__ __forceinline__ uint4 rw_pack8(const float (&v)[8]);
template <> __device__ __forceinline__ uint4 rw_pack8<bf16_t>(const float (&v)[8])
{
    return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
template <> __device__ __forceinline__ uint4 rw_pack8<f16_t>(const float (&v)[8])
{
    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
    h16x8 h;
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = (_Float16)v[k];
    return __builtin_bit_cast(uint4, h);
}

// ---- what a wave issues, in order, in the steady state; one entry per (tap kj, k-step q) of a stage:
//   TN weight loads of tap t + PF (fragment q of every channel tile), then nx(kj, q) pixel pieces of stage s + DX
//   (taps 1 and 2 only: piece j goes to k-step j % 8 of the stage's eight, first tap 1 / tap 2 alternating).
constexpr int rw_nx(int kj, int q, int CX)
{
    if (kj == 0) return 0;
    int n = 0;
    for (int j = 0; j < CX; ++j) {
        const int idx = j % 8;
        if (1 + (idx & 1) == kj && (idx >> 1) == q) ++n;
    }
    return n;
}
// N of the s_waitcnt vmcnt(N) in front of tap kj (its weights, loaded PF taps earlier) and of the one behind tap 0 (the
// pixel tile of the NEXT stage, issued DX - 1 stages earlier in taps 1 / 2): operations issued after the last one waited for.
struct RwWaits { int aw[3]; int ax; };
constexpr RwWaits rw_waits(int TN, int PF, int DX, int CX)
{
    // simulate stages 0 .. S-1; an operation's group: weights of tap u -> u, pixel tile of stage v -> 1000 + v
    constexpr int S = 12;
    int last_w[3 * S + 64] = {}, last_x[S + 16] = {};
    int pos_tap[3 * S] = {};               // operations issued before tap (s, kj) starts
    for (int i = 0; i < 3 * S + 64; ++i) last_w[i] = -1;
    for (int i = 0; i < S + 16; ++i) last_x[i] = -1;
    int n = 0;
    for (int s = 0; s < S; ++s)
        for (int kj = 0; kj < 3; ++kj) {
            pos_tap[3 * s + kj] = n;
            for (int q = 0; q < 4; ++q) {
                n += TN; last_w[3 * s + kj + PF] = n - 1;
                const int nx = rw_nx(kj, q, CX);
                if (nx) { n += nx; last_x[s + DX] = n - 1; }
            }
        }
    RwWaits r = {};
    const int s = S - 3;
    for (int kj = 0; kj < 3; ++kj) r.aw[kj] = pos_tap[3 * s + kj] - 1 - last_w[3 * s + kj];
    r.ax = last_x[s + 1] < 0 ? 63 : pos_tap[3 * s + 1] - 1 - last_x[s + 1];
    return r;
}

// Block = WN x WM waves.  Wave (wn, wm) owns channel tiles wn*TN .. +TN-1 (32 channels each) and its even share of the
// workgroup's npt position tiles (at most CMAX).  PF = taps the weight loads run ahead (PF + 1 register sets, PF + 1 = 3:
// set = tap of the stage), DX = stages the pixel DMA runs ahead (DX + 1 slots).
template <typename T, int TN, int CMAX, int WN, int WM, int PF, int DX>
__global__ void __launch_bounds__(WN * WM * 64) k_conv3x3_rw(RwArgs a)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    static_assert(PF == 2, "register sets rotate with the tap of the stage");
    constexpr int NW = WN * WM;
    constexpr int BN = WN * TN * 32;
    constexpr int BMMAX = WM * CMAX * 32;
    constexpr int PXW = (BMMAX + 2 + 8 * NW - 1) / (8 * NW);  // most pixel pieces a wave issues per stage
    static_assert(PXW <= 7, "the piece-count dispatch below enumerates 0 .. 7");
    constexpr int XROWS = (BMMAX + 2 + 7) / 8 * 8;            // a wave issues exactly its pieces below ceil((BM + 2) / 8): none past the slot
    constexpr int XSLOT = XROWS * 128;
    constexpr int NSX = DX + 1;
    static_assert(NSX * XSLOT <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char lds[NSX * XSLOT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Waves w and w + 4 share a SIMD (and its matrix pipe): the second half of the workgroup takes the position shares in
    // reverse order, so that a wave with one tile more is paired with a wave with one tile less.
    const int wn = wid / WM;
    const int wm = (NW == 8 && (wid & 4)) ? WM - 1 - wid % WM : wid % WM;
    const int r = lane & 31, h = lane >> 5;

    // PERSISTENT workgroups, XCD-aware tile order (speed only): XCD x = blockIdx & 7 owns the x-th contiguous chunk of the
    // (position tile, channel tile) list, channel tiles fastest; its workgroups take the chunk's tiles round-robin.
    const int nt = a.Cn / BN;
    const int nblk = a.mtiles * nt;
    const int chunk = (nblk + 7) >> 3;
    const int wpx = gridDim.x >> 3;                   // workgroups per XCD
    const int chunk_lo = (blockIdx.x & 7) * chunk, chunk_hi = min(chunk_lo + chunk, nblk);
    int gidx = chunk_lo + (blockIdx.x >> 3);
    if (gidx >= chunk_hi) return;
    int n0 = 0, q0 = 0;
    const int BM = a.npt * 32;
    const int Wp = a.W + 2, BH = a.B * a.H;
    const int rowbytes = a.Ck * 2;
    const int cchunks = rowbytes / 128;
    const int nstage = 3 * cchunks;

    // this wave's share of the position tiles
    const int base = a.npt / WM, rem = a.npt - base * WM;
    const int cnt = base + (wm < rem ? 1 : 0);
    const int pt0 = wm * base + min(wm, rem);

    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcW = __builtin_amdgcn_make_buffer_rsrc((void *)a.wf, 0, a.wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dstY = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.ybytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;            // pixel pieces / stores (no immediate offset)
    constexpr unsigned OOBW = 0x80000000u;           // weight loads (immediate offsets up to 3072 must not wrap)
    const unsigned ldsX0 = lds_addr(lds);

    // ---- pixel DMA.  Lane = (row l8 of the 8-row piece, 16-B chunk position lc); LDS position lc of row R holds source chunk
    // lc ^ ((R >> 1) & 7): the 16 rows of a ds_read_b128 lane group then sit on distinct banks (also at row offsets 1 and 2).
    const int l8 = lane >> 3, lc = lane & 7;
    const int npieces = (BM + 2 + 7) >> 3;
    const int cntx = __builtin_amdgcn_readfirstlane(wid < npieces ? (npieces - 1 - wid) / NW + 1 : 0);
    int xbase[PXW], xok[PXW];
    unsigned wtile[TN];                               // per-lane byte offset of the wave's channel tiles in the fragment image
    const int rowpitch = a.W * rowbytes;
    const unsigned tapbytes = (unsigned)cchunks * 4096u;      // one tap of one channel tile
    auto setup_tile = [&](int gi) __attribute__((always_inline)) {
        n0 = (gi % nt) * BN;
        q0 = (gi / nt) * BM;                               // first padded position of this tile
#pragma unroll
        for (int i = 0; i < TN; ++i) wtile[i] = (unsigned)(n0 / 32 + wn * TN + i) * 9u * tapbytes + (unsigned)lane * 16u;
#pragma unroll
        for (int j = 0; j < PXW; ++j) {
            const int i = (wid + j * NW) * 8 + l8;
            const int p = q0 - 1 + i;
            const int R = p >= 0 ? p / Wp : 0;
            const int c = p - R * Wp;
            const int oh = R % a.H;
            const bool live = (i < BM + 2) && (p >= 0) && (R < BH) && (c >= 1) && (c <= a.W);
            xbase[j] = (R * a.W + c - 1) * rowbytes + ((lc ^ ((i >> 1) & 7)) * 16);
            xok[j] = live ? ((oh >= 1 ? 1 : 0) | 2 | (oh + 1 < a.H ? 4 : 0)) : 0;
        }
    };
    setup_tile(gidx);
    f32x16 acc[TN][CMAX];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < CMAX; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // ---- pixel fragment reads.  k-step ks, lane half h reads source chunk 2 ks + h of its row, stored at position
    // (2 ks + h) ^ key(row): byte offset ((h ^ key) << 4) ^ (ks << 5), key = (row >> 1) & 7, row = r + kj (+ multiples of 32).
    const int rdX = (pt0 * 32 + r) * 128;

    // ---- epilogue of one tile (as k_conv3x3_rs): v = acc + shift + res ; relu ; v *= (mask > 0) ; 8 consecutive channels per access
    const T *res = reinterpret_cast<const T *>(a.res);
    const T *mask = reinterpret_cast<const T *>(a.mask);
    auto store_tile = [&](int q0c, int n0c) __attribute__((always_inline)) {
        if (RW_DBG & 4) {                                   // keep the accumulators alive: the MFMAs must not become dead code
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < CMAX; ++j) rw_keep(acc[i][j]);
            return;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < CMAX; ++j) acc_rows8(acc[i][j]);
        int mrow[CMAX];
        bool valid[CMAX];
#pragma unroll
        for (int j = 0; j < CMAX; ++j) {
            const int p = q0c + (pt0 + j) * 32 + r;
            const int R = p / Wp, c = p - R * Wp;
            valid[j] = (j < cnt) && !(p >= a.Q || c < 1 || c > a.W);      // padding position: no output
            mrow[j] = R * a.W + c - 1;
        }
        auto voff = [&](int j, int i, int pp) { return (size_t)mrow[j] * a.Cn + n0c + (wn * TN + i) * 32 + 16 * pp + 8 * h; };
        if (res) {
            uint4 rr[CMAX][TN][2];
#pragma unroll
            for (int j = 0; j < CMAX; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        rr[j][i][pp] = valid[j] ? *reinterpret_cast<const uint4 *>(res + voff(j, i, pp)) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < CMAX; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const unsigned rw[4] = {rr[j][i][pp].x, rr[j][i][pp].y, rr[j][i][pp].z, rr[j][i][pp].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo, hi;
                            unpack2<T>(rw[e], lo, hi);
                            acc[i][j][8 * pp + 2 * e] += lo; acc[i][j][8 * pp + 2 * e + 1] += hi;
                        }
                    }
        }
        if (a.shift) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    float sh[8];
                    ld8(a.shift + n0c + (wn * TN + i) * 32 + 16 * pp + 8 * h, sh);
#pragma unroll
                    for (int j = 0; j < CMAX; ++j)
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[i][j][8 * pp + k] += sh[k];
                }
        }
        if (a.relu) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[i][j][k] = fmaxf(acc[i][j][k], 0.f);
        }
        if (mask) {
            uint4 mm[CMAX][TN][2];
#pragma unroll
            for (int j = 0; j < CMAX; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        mm[j][i][pp] = valid[j] ? *reinterpret_cast<const uint4 *>(mask + voff(j, i, pp)) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < CMAX; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const unsigned mw[4] = {mm[j][i][pp].x, mm[j][i][pp].y, mm[j][i][pp].z, mm[j][i][pp].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo, hi;
                            unpack2<T>(mw[e], lo, hi);
                            if (!(lo > 0.f)) acc[i][j][8 * pp + 2 * e] = 0.f;
                            if (!(hi > 0.f)) acc[i][j][8 * pp + 2 * e + 1] = 0.f;
                        }
                    }
        }
#pragma unroll
        for (int j = 0; j < CMAX; ++j) {
            if (j >= cnt) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = acc[i][j][8 * pp + k];
                    rw_gst16(dstY, valid[j] ? (unsigned)(voff(j, i, pp) * sizeof(T)) : OOB, rw_pack8<T>(v));
                }
        }
    };

    // Main loop, specialised on the wave's tile count C and on its pixel-piece count CX (both wave-uniform).
    auto main_loop = [&](auto CNT, auto CNTX) {
        constexpr int C = decltype(CNT)::value, CX = decltype(CNTX)::value;
        constexpr int CR = C > 0 ? C : 1;
        constexpr RwWaits WT = rw_waits(TN, PF, DX, CX);
        static_assert(WT.aw[0] < 64 && WT.aw[1] < 64 && WT.aw[2] < 64 && WT.ax < 64 && WT.aw[0] >= 0 && WT.ax >= 0, "vmcnt range");
        constexpr int DS = (2 + PF) / 3 > DX ? (2 + PF) / 3 : DX;          // stages the bookkeeping looks ahead
        // stage coordinates of stages s .. s + DS: weight offset of the stage's tap 0 (relative to the channel tile), pixel
        // offset, kernel row (3 = past the end)
        int wst[DS + 1], xst[DS + 1], kis[DS + 1];
        int lki = 0, lcc = 0;
        auto stage_entry = [&](int d) __attribute__((always_inline)) {
            kis[d] = lki > 2 ? 3 : lki;
            wst[d] = ((a.flip ? 8 - 3 * lki : 3 * lki) * cchunks + lcc) * 4096;
            xst[d] = (lki - 1) * rowpitch + lcc * 128;
            if (++lcc == cchunks) { lcc = 0; ++lki; }
        };
        const int tapstep = a.flip ? -(int)tapbytes : (int)tapbytes;
        u32x4 wr[PF + 1][TN][4];                                           // weight fragments: [set][channel tile][k-step]
        if (RW_DBG & 64) {
#pragma unroll
            for (int u = 0; u <= PF; ++u)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { wr[u][i][q] = (u32x4){(unsigned)lane, 1u, 2u, 3u}; rw_bind(wr[u][i][q]); }
        }
        // fragment q of every channel tile of tap kw of stage s + d, into register set `set`
        auto issue_w = [&](int d, int kw, auto SET, auto QQ) __attribute__((always_inline)) {
            constexpr int set = decltype(SET)::value, q = decltype(QQ)::value;
            if (RW_DBG & 64) return;
            const bool ok = kis[d] < 3 && !(RW_DBG & 16);
            const unsigned koff = (unsigned)(wst[d] + kw * tapstep);
#pragma unroll
            for (int i = 0; i < TN; ++i) gldw<q * 1024>(wr[set][i][q], srcW, ok ? wtile[i] + koff : OOBW);
        };
        auto issue_x = [&](int d, int slot, int j) __attribute__((always_inline)) {       // piece j of the pixel tile of stage s + d
            if (RW_DBG & 32) return;
            const int ki = kis[d];
            const unsigned dst = __builtin_amdgcn_readfirstlane(ldsX0 + slot * XSLOT + (wid + j * NW) * 1024);
            glds16(srcX, (((xok[j] >> ki) & 1) && !(RW_DBG & 2)) ? (unsigned)(xbase[j] + xst[d]) : OOB, dst);
        };
        // the C pixel fragments of k-step q of tap kj out of slot `slot`
        auto read_x = [&](uint4 (&fb)[CR], int slot, int kj, int q) __attribute__((always_inline)) {
            if (RW_DBG & 128) return;
            const char *px = lds + slot * XSLOT + rdX + kj * 128;
            const int swx = ((h ^ (((r + kj) >> 1) & 7)) << 4) ^ (q << 5);
#pragma unroll
            for (int j = 0; j < C; ++j) fb[j] = *reinterpret_cast<const uint4 *>(px + j * 32 * 128 + swx);
        };
        auto bind_w = [&](auto SET) __attribute__((always_inline)) {      // uses of the set's registers stay behind the wait in front
            constexpr int set = decltype(SET)::value;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) rw_bind(wr[set][i][q]);
        };

        for (;;) {
            // ---- prologue of a tile: pixel tiles of stages 0 .. DX-1, weights of taps 0 .. PF-1; then everything landed
            lki = 0; lcc = 0;
#pragma unroll
            for (int d = 0; d <= DS; ++d) stage_entry(d);
#pragma unroll
            for (int sv = 0; sv < DX; ++sv)
#pragma unroll
                for (int j = 0; j < PXW; ++j)
                    if (j < CX) issue_x(sv, sv, j);
            // taps 0 .. PF-1 = taps 0, 1 of stage 0 (PF = 2)
            issue_w(0, 0, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
            issue_w(0, 0, std::integral_constant<int, 0>(), std::integral_constant<int, 1>());
            issue_w(0, 0, std::integral_constant<int, 0>(), std::integral_constant<int, 2>());
            issue_w(0, 0, std::integral_constant<int, 0>(), std::integral_constant<int, 3>());
            issue_w(0, 1, std::integral_constant<int, 1>(), std::integral_constant<int, 0>());
            issue_w(0, 1, std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
            issue_w(0, 1, std::integral_constant<int, 1>(), std::integral_constant<int, 2>());
            issue_w(0, 1, std::integral_constant<int, 1>(), std::integral_constant<int, 3>());
            wait_vmcnt<0>();
            bind_w(std::integral_constant<int, 0>());
            bind_w(std::integral_constant<int, 1>());
            __builtin_amdgcn_s_barrier();

            uint4 fbc[CR], fbn[CR];
            if (RW_DBG & 128) {
#pragma unroll
                for (int j = 0; j < CR; ++j) {
                    u32x4 v = {(unsigned)lane, 5u, 6u, 7u};
                    rw_bind(v);
                    fbc[j] = fbn[j] = __builtin_bit_cast(uint4, v);
                }
            }
            int xsr = 0, xsi = DX % NSX;                                   // ring slots: read / issue
            read_x(fbc, 0, 0, 0);
            for (int s = 0; s < nstage; ++s) {
                const int xsn = xsr + 1 == NSX ? 0 : xsr + 1;             // slot of stage s + 1
                auto tap = [&](auto KJ) __attribute__((always_inline)) {
                    constexpr int kj = decltype(KJ)::value;
                    constexpr int setc = kj;                               // PF + 1 = 3 sets: set = tap of the stage
                    constexpr int seti = (kj + PF) % 3;                    // = set of tap t + PF
                    constexpr int dwi = (kj + PF) / 3, kwi = (kj + PF) % 3;
                    wait_vmcnt_nomem<WT.aw[kj]>();
                    bind_w(std::integral_constant<int, setc>());
                    auto quarter = [&](auto QQ) __attribute__((always_inline)) {
                        constexpr int q = decltype(QQ)::value;
                        // fragments of the next k-step (next tap / next stage at the ends) before this one's MFMAs
                        if constexpr (C > 0) {
                            if constexpr (q < 3) read_x(fbn, xsr, kj, q + 1);
                            else if constexpr (kj < 2) read_x(fbn, xsr, kj + 1, 0);
                            else read_x(fbn, xsn, 0, 0);
                        }
                        // hipcc's scheduler otherwise sinks these reads down to their uses (one k-step later): the reads stay
                        // here, a whole MFMA group ahead of the MFMAs that consume them
                        __builtin_amdgcn_sched_barrier(0);
                        issue_w(dwi, kwi, std::integral_constant<int, seti>(), QQ);
#pragma unroll
                        for (int j = 0; j < PXW; ++j)
                            if (kj > 0 && j < CX && 1 + ((j % 8) & 1) == kj && ((j % 8) >> 1) == q) issue_x(DX, xsi, j);
                        if constexpr (C > 0) {
                            if (!(RW_DBG & 1)) {
#pragma unroll
                                for (int j = 0; j < C; ++j)
#pragma unroll
                                    for (int i = 0; i < TN; ++i)
                                        Mma<T>::run(__builtin_bit_cast(uint4, wr[setc][i][q]), fbc[j], acc[i][j]);
                            }
#pragma unroll
                            for (int j = 0; j < C; ++j) fbc[j] = fbn[j];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    quarter(std::integral_constant<int, 0>());
                    quarter(std::integral_constant<int, 1>());
                    quarter(std::integral_constant<int, 2>());
                    quarter(std::integral_constant<int, 3>());
                    if constexpr (kj == 0) {
                        // pixel tile of stage s + 1 landed (this wave's pieces), everyone has left stage s - 1
                        wait_vmcnt<WT.ax>();
                        __builtin_amdgcn_s_barrier();
                    }
                };
                tap(std::integral_constant<int, 0>());
                tap(std::integral_constant<int, 1>());
                tap(std::integral_constant<int, 2>());
                xsr = xsn;
                xsi = xsi + 1 == NSX ? 0 : xsi + 1;
#pragma unroll
                for (int d = 0; d < DS; ++d) { wst[d] = wst[d + 1]; xst[d] = xst[d + 1]; kis[d] = kis[d + 1]; }
                stage_entry(DS);
            }
            wait_vmcnt<0>();              // the trailing (out-of-range) loads and pieces still target this wave's registers / LDS
            bind_w(std::integral_constant<int, 0>());
            bind_w(std::integral_constant<int, 1>());
            bind_w(std::integral_constant<int, 2>());
            const int q0c = q0, n0c = n0;
            gidx += wpx;
            const bool more = gidx < chunk_hi;
            store_tile(q0c, n0c);
            if (!more) break;
            __builtin_amdgcn_s_barrier();                                  // every wave is done with the ring
            setup_tile(gidx);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        }
    };
    // dispatch on (tiles, pixel pieces) of this wave; the host's plan keeps both inside the instantiated ranges
#define DCF_RW_CX(C_)                                                                                                       \
    switch (cntx) {                                                                                                          \
    case 0: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, 0>()); break;                           \
    case 1: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 1 ? 1 : PXW)>()); break;        \
    case 2: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 2 ? 2 : PXW)>()); break;        \
    case 3: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 3 ? 3 : PXW)>()); break;        \
    case 4: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 4 ? 4 : PXW)>()); break;        \
    case 5: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 5 ? 5 : PXW)>()); break;        \
    case 6: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 6 ? 6 : PXW)>()); break;        \
    default: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, PXW>()); break;                        \
    }
    switch (cnt) {
    case 0: DCF_RW_CX(0) break;
    case 1: DCF_RW_CX(1) break;
    case 2: DCF_RW_CX((CMAX >= 2 ? 2 : CMAX)) break;
    case 3: DCF_RW_CX((CMAX >= 3 ? 3 : CMAX)) break;
    case 4: DCF_RW_CX((CMAX >= 4 ? 4 : CMAX)) break;
    default: DCF_RW_CX(CMAX) break;
    }
#undef DCF_RW_CX
}

// [Cn][9][Ck] -> fragment order (see the head of the file).  One thread per 16-byte fragment piece.
template <typename T>
__global__ void __launch_bounds__(256) k_weight_frag(const uint4 *w, uint4 *wf, int Cn, int Ck, long n16)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= n16) return;
    const int cch = Ck / 64;
    const int lane = (int)(g & 63);
    long b = g >> 6;                        // block = ((ct * 9 + tap) * cch + cc) * 4 + ks
    const int ks = (int)(b & 3); b >>= 2;
    const int cc = (int)(b % cch); b /= cch;
    const int tap = (int)(b % 9);
    const int ct = (int)(b / 9);
    const int r = lane & 31, hh = lane >> 5;
    const long src = (((long)(ct * 32 + r) * 9 + tap) * Ck + cc * 64 + ks * 16 + hh * 8) / 8;
    wf[g] = w[src];
}

// Tile shape of a launch: kind 0 = 128 channels x up to 320 positions (waves 4 x 2, up to 5 position tiles per wave),
// kind 1 = 64 channels x up to 384 positions (waves 2 x 4, up to 3 per wave).
struct RwPlan { int kind, npt; };
struct RwKind { int BN, WM, CMAX; };
static const RwKind RW_KINDS[2] = {{128, 2, 5}, {64, 4, 3}};

static RwPlan rw_plan(int64_t Q, int Cn)
{
    static DcfOpt ek_o("RW_KIND"), en_o("RW_NPT");
    const char *ek = ek_o.str(), *en = en_o.str();
    const int ncu = 256;
    RwPlan best = {-1, 0};
    double best_t = 1e30;
    for (int kind = 0; kind < 2; ++kind) {
        const RwKind &k = RW_KINDS[kind];
        if (Cn % k.BN) continue;
        if (ek && atoi(ek) != kind) continue;
        for (int npt = 1; npt <= k.WM * k.CMAX; ++npt) {
            if (en && atoi(en) != npt) continue;
            const int64_t tiles = (Q + 32 * npt - 1) / (32 * npt) * (Cn / k.BN);
            const int64_t rounds = (tiles + ncu - 1) / ncu;
            const int per_wave = (npt + k.WM - 1) / k.WM;
            // cycles per tap on a CU: MFMAs of the busiest SIMD (2 waves), a stage's barrier shared by its three taps, the
            // weight fragments (L2 -> registers, 8 waves x 4 KiB per tap at ~48 B/clk), and a fixed cost per tile
            const double mfma = 2.0 * 4 * per_wave * 32;
            const double wld = 8.0 * 4096.0 / 48.0;
            const double step = std::max(mfma, wld) + 100.0;
            const double t = rounds * (step + 60.0 * per_wave /* prologue + epilogue share */);
            if (t < best_t) { best_t = t; best = {kind, npt}; }
        }
    }
    return best;
}

}  // namespace

// Called by dcf_conv3x3_fwd_wf / dcf_conv3x3_dgrad_wf (below).  Returns DCF_EUNSUPPORTED when the shape is not this kernel's.
static int conv3x3_rw_launch(int dtype, const void *x, const void *wf, const float *shift, const void *res, const void *mask, void *y,
                             int B, int H, int W, int Ck, int Cn, int relu, int flip, const char *name_base, hipStream_t s)
{
    if ((dtype != DCF_BF16 && dtype != DCF_F16) || Ck % 64 || Cn % 64 || Ck < 64 || Cn < 64) return DCF_EUNSUPPORTED;
    const int64_t Q = (int64_t)B * H * (W + 2);
    if (Q >= (1ll << 30) || (int64_t)B * H * W * Ck * 2 >= (1ll << 31) || (int64_t)B * H * W * Cn * 2 >= 0xFFFFFF00ll ||
        (int64_t)Cn * 9 * Ck * 2 >= (1ll << 31))
        return DCF_EUNSUPPORTED;
    const RwPlan p = rw_plan(Q, Cn);
    if (p.kind < 0) return DCF_EUNSUPPORTED;
    RwArgs a;
    a.x = (const char *)x; a.wf = (const char *)wf; a.shift = shift; a.res = (const char *)res; a.mask = (const char *)mask; a.y = (char *)y;
    a.B = B; a.H = H; a.W = W; a.Ck = Ck; a.Cn = Cn; a.relu = relu; a.flip = flip;
    a.npt = p.npt; a.Q = (int)Q;
    a.mtiles = (int)((Q + 32 * p.npt - 1) / (32 * p.npt));
    a.xbytes = (unsigned)((int64_t)B * H * W * Ck * 2);
    a.wbytes = (unsigned)((int64_t)Cn * 9 * Ck * 2);
    a.ybytes = (unsigned)((int64_t)B * H * W * Cn * 2);
    const int BN = RW_KINDS[p.kind].BN;
    int64_t nwg = (((int64_t)a.mtiles * (Cn / BN) + 7) / 8) * 8;
    nwg = std::min<int64_t>(nwg, 256);                       // persistent workgroups: at most one per CU
    const dim3 grid((unsigned)nwg);
    char name[96];
    snprintf(name, sizeof(name), "%s<rw%d,%d>", name_base, p.kind, p.npt);
    const double flops = 2.0 * B * H * W * (double)Cn * Ck * 9.0;
    const double bytes = (double)a.xbytes + (double)a.wbytes + (double)B * H * W * Cn * 2.0 * (1 + (res ? 1 : 0) + (mask ? 1 : 0));
#define DCF_RW(T_)                                                                                                               \
    do {                                                                                                                         \
        if (p.kind == 0) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rw<T_, 1, 5, 4, 2, 2, 2>), grid, dim3(512), 0, s, a)); \
        else DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rw<T_, 1, 3, 2, 4, 2, 2>), grid, dim3(512), 0, s, a)); \
    } while (0)
#ifdef RW_BF16_ONLY            /* tools/rw_variants.sh: half the compile time */
    if (dtype == DCF_F16) return DCF_EUNSUPPORTED;
    DCF_RW(bf16_t);
#else
    if (dtype == DCF_F16) DCF_RW(f16_t); else DCF_RW(bf16_t);
#endif
#undef DCF_RW
    return DCF_OK;
}

// ================================================================== C ABI
extern "C" int dcf_conv3x3_weight_frag(int dtype, const void *w, void *wf, int Cout, int Cin, dcf_stream_t stream)
{
    DCF_REQUIRE(dtype == DCF_BF16 || dtype == DCF_F16, "dcf_conv3x3_weight_frag: 16-bit dtypes only (got %d)", dtype);
    DCF_REQUIRE(w && wf && w != wf, "dcf_conv3x3_weight_frag: null or aliased pointer");
    DCF_REQUIRE(Cout > 0 && Cin > 0 && Cout % 32 == 0 && Cin % 64 == 0, "dcf_conv3x3_weight_frag: Cout %% 32 and Cin %% 64 must be 0 (%d, %d)", Cout, Cin);
    const long n16 = (long)Cout * 9 * Cin / 8;
    DCF_LAUNCH_B("weight_frag", (double)n16 * 32.0, S(stream),
                 hipLaunchKernelGGL(k_weight_frag<bf16_t>, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, S(stream),
                                    (const uint4 *)w, (uint4 *)wf, Cout, Cin, n16));
    return DCF_OK;
}

extern "C" int dcf_conv3x3_wf_supported(int dtype, int B, int H, int W, int Cin, int Cout)
{
    if ((dtype != DCF_BF16 && dtype != DCF_F16) || Cin % 64 || Cout % 64 || Cin < 64 || Cout < 64 || B < 1 || H < 1 || W < 1) return 0;
    const int64_t Q = (int64_t)B * H * (W + 2);
    const int64_t cmax = std::max(Cin, Cout);
    if (Q >= (1ll << 30) || (int64_t)B * H * W * cmax * 2 >= (1ll << 31) || (int64_t)Cout * 9 * Cin * 2 >= (1ll << 31)) return 0;
    return 1;
}

extern "C" int dcf_conv3x3_fwd_wf(int dtype, const void *x, const void *wf, const float *shift, const void *res, void *y,
                                  int B, int H, int W, int Cin, int Cout, int relu, dcf_stream_t stream)
{
    DCF_REQUIRE(x && wf && y, "dcf_conv3x3_fwd_wf: null pointer");
    DCF_REQUIRE(dcf_conv3x3_wf_supported(dtype, B, H, W, Cin, Cout), "dcf_conv3x3_fwd_wf: unsupported shape / dtype (%d: %dx%dx%d, %d -> %d)", dtype, B, H, W, Cin, Cout);
    return conv3x3_rw_launch(dtype, x, wf, shift, res, nullptr, y, B, H, W, Cin, Cout, relu, 0, dtype == DCF_F16 ? "conv_fwd_f16" : "conv_fwd_bf16", S(stream));
}

extern "C" int dcf_conv3x3_dgrad_wf(int dtype, const void *gy, const void *wtf, const void *res, const void *mask, void *gx,
                                    int B, int H, int W, int Cin, int Cout, dcf_stream_t stream)
{
    DCF_REQUIRE(gy && wtf && gx, "dcf_conv3x3_dgrad_wf: null pointer");
    DCF_REQUIRE(dcf_conv3x3_wf_supported(dtype, B, H, W, Cin, Cout), "dcf_conv3x3_dgrad_wf: unsupported shape / dtype (%d: %dx%dx%d, %d -> %d)", dtype, B, H, W, Cin, Cout);
    // roles swap: reduction channels = Cout, produced channels = Cin
    return conv3x3_rw_launch(dtype, gy, wtf, nullptr, res, mask, gx, B, H, W, Cout, Cin, 0, 1, dtype == DCF_F16 ? "conv_dgrad_f16" : "conv_dgrad_bf16", S(stream));
}
