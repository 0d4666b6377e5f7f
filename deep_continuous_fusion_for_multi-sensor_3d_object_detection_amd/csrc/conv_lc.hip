// conv_lc.hip -- implicit GEMM with LOADER and CONSUMER waves over 2-D SPATIAL TILES, for the 3x3 / stride-1 / pad-1
// convolutions (forward and input gradient), 16-bit element types, gfx950.
//
// The layers are those of conv_rs.hip (/root/reference/model.py:15-28 ResidualBlock bodies, :153 conv3; the camera trunk's
// BasicBlocks).  What round 4's measurements on conv_rs.hip / conv_rw.hip and on a first, row-sharing version of this file said
// (profiles/r04a_sq_*.csv, r04b_rw_ablation.txt, r04c_lc1d_*.txt, r04c_probe_*.txt):
//   * the MFMA + fragment-read stream alone runs at the matrix pipes' pace (1.96 PFLOP/s on the 128-channel stage);
//   * a CU takes one vector-memory wave-instruction per ~16 cycles at best, and a wave that issues LDS-DMA pieces between its
//     MFMAs stalls them (conv_rs: 60 % of wave cycles parked, conv_rw: 26 % issue-stalled);
//   * weights (the same 16 KiB per tap for every workgroup, L2-resident) stream for free; what the loop waits for is the PIXEL
//     tiles: the row-sharing scheme stages every pixel three times (once per kernel row), ~13 KiB-pieces per tap, and the
//     waves that issue them are the last to reach every barrier.
// Hence:
//   * 2-D TILES WITH A HALO.  A workgroup owns TH x TW output pixels of one frame (TH * TW <= 320) x BN channels.  Per
//     64-channel chunk it stages the (TH + 2) x (TW + 2) input pixels ONCE (image borders = out-of-range DMA = zeros); all nine
//     taps read that tile: tap (ki, kj) of output (y, x) is staged row (y + ki)(TW + 2) + x + kj.  1.3-1.5 x the tile's pixels
//     instead of 3 x, and a staged slot lives for nine taps, so the next chunk's fill has ~5 us to land.
//   * waves 0-3 (one per SIMD) are CONSUMERS: fragment reads one k-step ahead + MFMAs, no vector-memory instruction in the loop,
//     a 2 x 2 arrangement of (TN x 32 channels) x (C x 32 positions) register tiles; positions are the tile's pixels in
//     row-major order, 32 per MFMA tile (a lane computes its pixel's staged row once per kernel);
//   * waves 4-5 are WEIGHT LOADERS (a ring of NSW one-tap slots of the [Cn][9][Ck] image, two taps ahead), waves 6-7 PIXEL
//     LOADERS (two chunk slots).  Two kinds of loader because s_waitcnt vmcnt retires in issue order: a wave that loaded both
//     would wait for its pixel pieces whenever it waits for a tap's weights.  Their code is straight-line (a first version that
//     branched around pieces spent ~190 cycles per piece in taken branches);
//   * the rings run across the tile boundaries of the persistent workgroup: the loaders fill the next tile's first slots while
//     the consumers store the current tile, whose stores drain under the next tile's MFMAs (no vmcnt wait in a consumer's loop);
//   * one s_barrier per tap, joined by all eight waves, between k-steps 2 and 3: barrier g + 1 (inside tap g) tells the
//     consumers that tap g + 1's weights (and at a chunk's last tap the next chunk's pixels) have landed -- the loaders wait
//     for their own pieces first -- and tells the loaders that tap g's weight slot (the chunk's pixel slot) is free.
//
// K order: chunk, tap, k-step (conv_rs.hip: kernel row, chunk, tap): same products, another fp32 summation order.
// dgrad = the same kernel on the [Cin][tap][Cout] weight image with the taps mirrored.
// Algorithmic work per launch: 2*B*H*W*Cout*Cin*9 flop; bytes B*H*W*(Cin + Cout)*2 + weights (+ residual / mask reads).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "dcf_common.h"
#include "conv_common.h"

// Timing ablations, COMPILE-TIME only (tools/rw_variants.sh with KFILE=conv_lc builds one library per mask): 1 no MFMAs, 2 pixel
// DMA reads nothing, 4 no epilogue, 16 weight DMA reads nothing, 128 no LDS fragment reads.  Results are wrong in those builds;
// the shipped library is built with 0.
#ifndef LC_DBG
#define LC_DBG 0
#endif

// In-kernel time stamps (tools/lc_stamps.py; -DLC_STAMP builds only): every wave of workgroups 0 .. LC_STAMP_WGS-1 records
// s_memtime when it ARRIVES at barrier g and when it LEAVES it -- who waits for whom, tap by tap.  The stamps go to a buffer of
// their own that nothing else reads.
#ifdef LC_STAMP
#define LC_STAMP_WGS 4
#define LC_STAMP_MAXG 160
__device__ long long g_lc_stamps[LC_STAMP_WGS][8][LC_STAMP_MAXG][2];
#define LC_BARRIER(gg)                                                                                                     \
    do {                                                                                                                    \
        const int g__ = (gg);                                                                                               \
        const bool st__ = blockIdx.x < LC_STAMP_WGS && g__ < LC_STAMP_MAXG && (threadIdx.x & 63) == 0;                      \
        if (st__) g_lc_stamps[blockIdx.x][wid][g__][0] = __builtin_amdgcn_s_memtime();                                      \
        __builtin_amdgcn_s_barrier();                                                                                       \
        if (st__) g_lc_stamps[blockIdx.x][wid][g__][1] = __builtin_amdgcn_s_memtime();                                      \
    } while (0)
#define LC_T(k)                                                                                                            \
    do {                                                                                                                    \
        if (blockIdx.x < LC_STAMP_WGS && (threadIdx.x & 63) == 0) g_lc_stamps[blockIdx.x][wid][LC_STAMP_MAXG - 8 + (k)][0] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LC_BARRIER(gg) __builtin_amdgcn_s_barrier()
#define LC_T(k) do { } while (0)
#endif

namespace {

struct LcArgs {
    const char *x;        // [B][H][W][Ck]
    const char *w;        // [Cn][9][Ck]
    const float *shift;   // [Cn] or null
    const char *res;      // [B*H*W][Cn] or null
    const char *mask;     // [B*H*W][Cn] or null: output *= (mask > 0)
    char *y;              // [B*H*W][Cn]
    int B, H, W, Ck, Cn;
    int relu, flip;       // flip = 1: input gradient (taps mirrored)
    int TH, TW;           // output pixels of a workgroup tile (TW even)
    int nty, ntx;         // tiles per frame
    int mtiles;           // spatial tiles of the launch = B * nty * ntx
    unsigned xbytes, wbytes, ybytes;
};

typedef unsigned lc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lc_gst16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, uint4 data)
{
    const lc_u32x4 d = {data.x, data.y, data.z, data.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)voff, 0, 0);
}
__device__ __forceinline__ uint4 lc_gld16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff)
{
    const lc_u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);       // out of range: zeros
    return make_uint4(d[0], d[1], d[2], d[3]);
}
__device__ __forceinline__ void lc_keep(const f32x16 &v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void lc_opaque(uint4 &v)
{
    lc_u32x4 t = {v.x, v.y, v.z, v.w};
    asm volatile("" : "+v"(t));
    v = make_uint4(t[0], t[1], t[2], t[3]);
}
// LDS-DMA piece without the M0 save / restore of glds16 (M0 is declared clobbered instead)
__device__ __forceinline__ void lc_dma(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
template <typename T> __device__ __forceinline__ uint4 lc_pack8(const float (&v)[8]);
template <> __device__ __forceinline__ uint4 lc_pack8<bf16_t>(const float (&v)[8])
{
    return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
template <> __device__ __forceinline__ uint4 lc_pack8<f16_t>(const float (&v)[8])
{
    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
    h16x8 h;
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = (_Float16)v[k];
    return __builtin_bit_cast(uint4, h);
}

// Consumers: WN x WM = 4 waves; consumer (wn, wm) owns channel tiles wn*TN .. +TN-1 (32 channels each) and its even share of
// the tile's 32-position groups (at most CMAX).  NSW weight slots (taps), two pixel slots (chunks) of XROWS staged pixels.
template <typename T, int TN, int CMAX, int WN, int WM, int NSW, int XROWS>
__global__ void __launch_bounds__(512) k_conv3x3_lc(LcArgs a)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    static_assert(WN * WM == 4, "one consumer per SIMD");
    static_assert(NSW >= 3 && XROWS % 8 == 0, "ring");
    constexpr int BN = WN * TN * 32;
    constexpr int WSLOT = BN * 128;
    constexpr int XSLOT = XROWS * 128;
    constexpr int NWL = BN / 8 / 2;                           // weight pieces (8 rows x 128 B) per weight loader and tap
    constexpr int PXL = (XROWS / 8 + 1) / 2;                  // most pixel pieces per pixel loader and chunk
    constexpr int PXI = 5;                                    // pixel pieces a pixel loader issues per tap
    constexpr int NIT = (PXL + PXI - 1) / PXI;                // taps a fill's issue is spread over
    static_assert(NIT <= 7, "a fill must be out at least two taps before the chunk's nine are over");
    static_assert(NWL >= 1 && (NSW - 2) * NWL < 64, "vmcnt range");
    // + 1 KiB that the pixel loaders' surplus pieces are written to
    static_assert(NSW * WSLOT + 2 * XSLOT + 1024 + 64 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char lds[NSW * WSLOT + 2 * XSLOT + 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    // ROLES BY SIMD.  The design wants one consumer per SIMD (each owns a matrix pipe) with one loader beside it, but which
    // SIMD a wave lands on is the hardware's choice -- with roles by wave number two consumers regularly shared a SIMD (in-kernel
    // stamps: those two ran every tap at half the pace of the other two, and everybody waited for them at every barrier).  So
    // every wave reads its SIMD id (HW_REG_HW_ID bits 5:4), the waves rank themselves within their SIMD, and the four
    // lowest-ranked waves in (rank, SIMD) order become the consumers: one per SIMD whenever the placement allows it.
    __shared__ int s_simd[8];
    const int hwid = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const int simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;
        if (lane == 0) s_simd[hwid] = simd;
    }
    __syncthreads();
    int wid;
    {
        int key[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            int rank = 0;
#pragma unroll
            for (int v = 0; v < 8; ++v) rank += (v < w && s_simd[v] == s_simd[w]) ? 1 : 0;
            key[w] = rank * 64 + s_simd[w] * 8 + w;
        }
        int mine = 0, role = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) mine = (w == hwid) ? key[w] : mine;
#pragma unroll
        for (int w = 0; w < 8; ++w) role += key[w] < mine ? 1 : 0;
        wid = __builtin_amdgcn_readfirstlane(role);           // 0-3 consumers, 4-5 weight loaders, 6-7 pixel loaders
    }

    // PERSISTENT workgroups, XCD-aware tile order (speed only): XCD x = blockIdx & 7 owns the x-th contiguous chunk of the
    // (spatial tile, channel tile) list, channel tiles fastest; its workgroups take the chunk's tiles round-robin.
    const int nt = a.Cn / BN;
    const int nblk = a.mtiles * nt;
    const int chunk = (nblk + 7) >> 3;
    const int wpx = gridDim.x >> 3;                   // workgroups per XCD
    const int chunk_lo = (blockIdx.x & 7) * chunk, chunk_hi = min(chunk_lo + chunk, nblk);
    const int gidx0 = chunk_lo + (blockIdx.x >> 3);
    if (gidx0 >= chunk_hi) return;
    const int ntile_wg = (chunk_hi - gidx0 + wpx - 1) / wpx;
    const int PW = a.TW + 2;                          // staged pixels per staged row (even)
    const int SR = (a.TH + 2) * PW;                   // staged pixels per chunk slot
    const int rowbytes = a.Ck * 2;
    const int cchunks = rowbytes / 128;
    const int ntaps = 9 * cchunks;
    const int G = ntile_wg * ntaps;                   // taps of this workgroup = barriers after the first
    constexpr unsigned OOB = 0x80000000u;             // + a chunk offset stays out of range without wrapping (tensors < 2 GiB)
    const unsigned ldsW0 = lds_addr(lds), ldsX0 = ldsW0 + NSW * WSLOT;
    const int l8 = lane >> 3, lc = lane & 7;          // DMA lane = (row of the 8-row piece, 16-byte chunk position)

    if (wid >= 6) {
        // ================================================================ PIXEL LOADER (lx = 0, 1: pieces lx, lx + 2, ...)
        // LDS row s of a slot = staged pixel (yq, xq) = (s / PW, s % PW); position lc of the row holds source chunk lc ^ key,
        // key = ((yq * TW + xq) >> 1) & 7: for a tap, the lanes of a position group read pixels whose yq * TW + xq are CONSECUTIVE
        // (also across the end of a tile row, where the LDS row jumps by the two halo columns), so the 16 lanes of a
        // ds_read_b128 group sit on distinct banks.  (A key from the LDS row itself left 26 % of the LDS cycles to bank
        // conflicts on a 6 x 50 tile, and the consumers that own the row ends ran every tap 1.6 x slower than the others.)
        const int lx = wid - 6;
        const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
        const int npieces = (SR + 7) >> 3;
        const int cntx = __builtin_amdgcn_readfirstlane(lx < npieces ? (npieces - 1 - lx) / 2 + 1 : 0);
        const unsigned ldsDump = ldsX0 + 2 * XSLOT;
        unsigned xoff[PXL];                                    // per piece: this lane's source offset of chunk 0, or OOB
        auto setup_x = [&](int gi) __attribute__((always_inline)) {
            const int mt = gi / nt;
            const int tx = mt % a.ntx, t2 = mt / a.ntx;
            const int ty = t2 % a.nty, b = t2 / a.nty;
            const int iy0 = ty * a.TH - 1, ix0 = tx * a.TW - 1;
            // LANE-PARALLEL: the DMA layout (lane = row l8 of the piece, 16-byte position lc) would have every lane walk all of
            // the wave's pieces one after the other (a first version: ~500 cycles per piece, 14 000 per tile in this ONE wave,
            // while the whole workgroup waited for its first fill).  Instead lane L works out (piece 8 k + L / 8, row L % 8) for
            // k = 0 .. PXL / 8: eight pieces per pass, and the DMA lanes fetch their row's result with a shuffle.  The result is
            // packed as offset | key << 4 (offsets are multiples of 128), so the DMA lane's source offset is packed ^ (lc << 4).
            const int jj = lane >> 3, l8s = lane & 7;
            const float rpw = 1.0f / (float)PW;
            unsigned packed[(PXL + 7) / 8];
#pragma unroll
            for (int k = 0; k < (PXL + 7) / 8; ++k) {
                const int sp = (lx + 2 * (8 * k + jj)) * 8 + l8s;                // staged pixel of (piece, row)
                const int yq = (int)(((float)sp + 0.5f) * rpw);                   // sp / PW (exact: sp < 2^12, PW < 2^8)
                const int xq = sp - yq * PW;
                const int iy = iy0 + yq, ix = ix0 + xq;
                const bool live = (sp < SR) && (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
                const unsigned key = (unsigned)(((yq * a.TW + xq) >> 1) & 7);    // swizzle key source: yq * TW + xq (see above)
                packed[k] = live ? (unsigned)(((b * a.H + iy) * a.W + ix) * rowbytes) | (key << 4) : OOB;
            }
#pragma unroll
            for (int j = 0; j < PXL; ++j)
                xoff[j] = (unsigned)__shfl((int)packed[j / 8], (j % 8) * 8 + l8, 64) ^ ((unsigned)lc << 4);
        };
        int gi = gidx0, cc = 0;                                // fill cursor: tile, chunk
        bool live = true;
        LC_T(0);
        setup_x(gi);
        LC_T(1);
        // PXI pieces of the cursor's chunk, straight-line: pieces this wave does not have (j >= cntx) go out of range into the
        // spare KiB.  IT = which PXI of the wave's pieces (a compile-time index: the offsets live in registers).
        auto issue_it = [&](auto IT, int slot) __attribute__((always_inline)) {
            constexpr int it = decltype(IT)::value;
            const unsigned cco = (unsigned)cc * 128u;
#pragma unroll
            for (int k = 0; k < PXI; ++k) {
                const int j = it * PXI + k;
                if (j < PXL) {
                    const bool real = j < cntx;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(real ? ldsX0 + slot * XSLOT + (lx + 2 * j) * 1024 : ldsDump);
                    lc_dma(srcX, (LC_DBG & 2) ? OOB : xoff[j] + cco, dst);
                }
            }
        };
        auto issue_dyn = [&](int it, int slot) __attribute__((always_inline)) {
            switch (it) {
            case 0: issue_it(std::integral_constant<int, 0>(), slot); break;
            case 1: issue_it(std::integral_constant<int, 1>(), slot); break;
            case 2: issue_it(std::integral_constant<int, 2>(), slot); break;
            case 3: issue_it(std::integral_constant<int, 3>(), slot); break;
            case 4: issue_it(std::integral_constant<int, 4>(), slot); break;
            case 5: issue_it(std::integral_constant<int, 5>(), slot); break;
            default: issue_it(std::integral_constant<int, 6>(), slot); break;
            }
        };
        auto advance = [&]() __attribute__((always_inline)) {
            if (++cc == cchunks) {
                cc = 0;
                gi += wpx;
                live = gi < chunk_hi;
                if (live) setup_x(gi);
            }
        };
        const int V = ntile_wg * cchunks;                      // chunks (fills) of this workgroup
        // fill 0 only: the consumers can start on it; fill 1 goes out during the first taps like every later fill
        for (int it = 0; it < NIT; ++it) issue_dyn(it, 0);
        LC_T(2);
        advance();
        wait_vmcnt<0>();
        LC_T(3);
        LC_BARRIER(0);                                         // barrier 0
        // Barrier 9 vq (inside the last tap of chunk vq - 1) certifies chunk vq, whose fill was issued during chunk vq - 1:
        // everything this wave has issued must have landed.  Behind it the slot of chunk vq - 1 is free: fill vq + 1, PXI
        // pieces per tap.
        int it = V > 1 ? 0 : -1, slot = 1;
        for (int g = 0; g < G; ++g) {
            const bool edge = (g + 1) % 9 == 0;
            if (edge) wait_vmcnt<0>();
            LC_BARRIER(g + 1);                                 // barrier g + 1
            if (edge && (g + 1) / 9 + 1 < V) it = 0;           // fill vq + 1 into the slot chunk vq - 1 had: slots alternate
            if (it >= 0) {
                issue_dyn(it, slot);
                if (++it == NIT) { it = -1; advance(); slot ^= 1; }
            }
        }
        wait_vmcnt<0>();
        return;
    }
    if (wid >= 4) {
        // ================================================================ WEIGHT LOADER (lw = 0, 1: pieces lw, lw + 2, ...)
        const int lw = wid - 4;
        const __amdgpu_buffer_rsrc_t srcW = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.wbytes, 0x00020000);
        unsigned wbase[NWL];
        auto setup_w = [&](int gi) __attribute__((always_inline)) {
            const int n0 = (gi % nt) * BN;
#pragma unroll
            for (int j = 0; j < NWL; ++j) {
                const int row = (lw + 2 * j) * 8 + l8;
                wbase[j] = (unsigned)(n0 + row) * (unsigned)(9 * rowbytes) + (unsigned)((lc ^ ((row >> 1) & 7)) * 16);
            }
        };
        int gi = gidx0, cc = 0, t9 = 0, slot = 0;              // fill cursor: tile, chunk, tap; ring slot
        bool live = true;
        setup_w(gi);
        auto issue_w = [&]() __attribute__((always_inline)) {
            if (!live) return;
            const int tapidx = a.flip ? 8 - t9 : t9;
            const unsigned koff = (unsigned)(tapidx * rowbytes + cc * 128);
            const unsigned dst = __builtin_amdgcn_readfirstlane(ldsW0 + slot * WSLOT + lw * 1024);
#pragma unroll
            for (int j = 0; j < NWL; ++j) lc_dma(srcW, (LC_DBG & 16) ? OOB : wbase[j] + koff, dst + 2 * j * 1024);
            slot = slot + 1 == NSW ? 0 : slot + 1;
            if (++t9 == 9) {
                t9 = 0;
                if (++cc == cchunks) {
                    cc = 0;
                    gi += wpx;
                    live = gi < chunk_hi;
                    if (live) setup_w(gi);
                }
            }
        };
        for (int u = 0; u < NSW; ++u) issue_w();               // fills 0 .. NSW-1
        wait_vmcnt<(NSW - 1) * NWL>();                         // fill 0 has landed
        LC_BARRIER(0);                                         // barrier 0
        // Barrier g + 1 certifies tap g + 1 (fill g + 1): behind it this wave has issued fills g + 2 .. g + NSW - 1.  After the
        // barrier tap g's slot is free: fill g + NSW.
        for (int g = 0; g < G; ++g) {
            if (g + NSW - 1 < G) wait_vmcnt<(NSW - 2) * NWL>(); else wait_vmcnt<0>();
            LC_BARRIER(g + 1);                                 // barrier g + 1
            issue_w();
        }
        wait_vmcnt<0>();
        return;
    }

    // ==================================================================== CONSUMERS
#ifdef LC_SWAPWM
    const int wn = wid / WM, wm = WM - 1 - wid % WM;
#else
    const int wn = wid / WM, wm = wid % WM;
#endif
    const int npos = a.TH * a.TW;
    const int npt = (npos + 31) >> 5;                          // 32-position groups of a tile
    const int base = npt / WM, rem = npt - base * WM;
    const int cnt = base + (wm < rem ? 1 : 0);
    const int pt0 = wm * base + min(wm, rem);
    const __amdgpu_buffer_rsrc_t dstY = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.ybytes, 0x00020000);

    // this lane's pixel in each of its position groups: the staged row of tap (0, 0) = y * PW + x for tile-local (y, x); -1 - that
    // for a position past the tile's end (it reads row 0 and stores nothing)
    int brow[CMAX];
#pragma unroll
    for (int j = 0; j < CMAX; ++j) {
        const int p = (pt0 + j) * 32 + r;
        const bool ok = (j < cnt) && (p < npos);
        const int y = ok ? p / a.TW : 0;
        const int x = ok ? p - y * a.TW : 0;
        brow[j] = ok ? y * PW + x : -1;
    }
    // weight fragments: k-step q, lane half h reads source chunk 2 q + h of its row, stored at position (2 q + h) ^ key(row):
    // byte offset ((h ^ key) << 4) ^ (q << 5), key = (row >> 1) & 7
    const int swa0 = (h ^ ((r >> 1) & 7)) << 4;
    const int rdA = (wn * TN * 32 + r) * 128;

    f32x16 acc[TN][CMAX];
    const T *res = reinterpret_cast<const T *>(a.res);
    const T *mask = reinterpret_cast<const T *>(a.mask);
    // epilogue of one tile: v = acc + shift + res ; relu ; v *= (mask > 0) ; 8 consecutive channels per access
    auto store_tile = [&](int gi) __attribute__((always_inline)) {
        if (LC_DBG & 4) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < CMAX; ++j) lc_keep(acc[i][j]);
            return;
        }
        const int n0c = (gi % nt) * BN;
        const int mt = gi / nt;
        const int tx = mt % a.ntx, t2 = mt / a.ntx;
        const int ty = t2 % a.nty, b = t2 / a.nty;
        int mrow[CMAX];
        bool valid[CMAX];
#pragma unroll
        for (int j = 0; j < CMAX; ++j) {
            const int br = brow[j] < 0 ? 0 : brow[j];
            const int y = br / PW, x = br - y * PW;
            const int yy = ty * a.TH + y, xx = tx * a.TW + x;
            valid[j] = (brow[j] >= 0) && (yy < a.H) && (xx < a.W);
            mrow[j] = (b * a.H + yy) * a.W + xx;
        }
        // One channel tile at a time (64-channel consumers: with both tiles' residual vectors in flight beside 160 accumulators
        // the kernel spills), in phases of independent loads: the tile's residual vectors, (shift, ReLU), its mask vectors, its
        // stores.  32-bit buffer offsets, out of range for a position without an output: no 64-bit address per vector, no branches.
        const __amdgpu_buffer_rsrc_t srcR = __builtin_amdgcn_make_buffer_rsrc((void *)a.res, 0, a.ybytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t srcM = __builtin_amdgcn_make_buffer_rsrc((void *)a.mask, 0, a.ybytes, 0x00020000);
        unsigned vbase[CMAX];
#pragma unroll
        for (int j = 0; j < CMAX; ++j) vbase[j] = valid[j] ? (unsigned)(((unsigned)mrow[j] * (unsigned)a.Cn + n0c + wn * TN * 32 + 8 * h) * sizeof(T)) : OOB;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            auto voff = [&](int j, int pp) { return vbase[j] + (unsigned)((i * 32 + 16 * pp) * sizeof(T)); };
#pragma unroll
            for (int j = 0; j < CMAX; ++j) acc_rows8(acc[i][j]);
            if (res) {
                uint4 rr[CMAX][2];
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) rr[j][pp] = lc_gld16(srcR, voff(j, pp));
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const unsigned rw[4] = {rr[j][pp].x, rr[j][pp].y, rr[j][pp].z, rr[j][pp].w};
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
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[i][j][k] = fmaxf(acc[i][j][k], 0.f);
            }
            if (mask) {
                uint4 mm[CMAX][2];
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) mm[j][pp] = lc_gld16(srcM, voff(j, pp));
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const unsigned mw[4] = {mm[j][pp].x, mm[j][pp].y, mm[j][pp].z, mm[j][pp].w};
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
                for (int pp = 0; pp < 2; ++pp) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = acc[i][j][8 * pp + k];
                    lc_gst16(dstY, voff(j, pp), lc_pack8<T>(v));
                }
            }
        }
    };

    // Main loop, specialised on the wave's group count C (wave-uniform).
    auto main_loop = [&](auto CNT) {
        constexpr int C = decltype(CNT)::value;
        constexpr int CR = C > 0 ? C : 1;
        // Fragments: the TN weight fragments of a k-step are double-buffered (all the step's MFMAs use them); a pixel fragment is
        // used by ONE pair of MFMAs and is refreshed IN PLACE right behind them with the next k-step's -- a whole MFMA group
        // before its next use.  (A first version read all of the next k-step's fragments ahead of the group: with five position
        // groups that is 14 LDS instructions in flight per wave, and those waves ran every tap at HALF the pace of waves with
        // four groups / 12 in flight -- MFMAs alone 1445 cycles per tap, with the reads 2900 -- the per-wave LDS queue was full.)
        uint4 fac[TN], fan[TN], fb[CR];
        if (LC_DBG & 128) {
#pragma unroll
            for (int i = 0; i < TN; ++i) { fac[i] = make_uint4(lane, 1, 2, 3); lc_opaque(fac[i]); fan[i] = fac[i]; }
#pragma unroll
            for (int j = 0; j < CR; ++j) { fb[j] = make_uint4(lane, 5, 6, 7); lc_opaque(fb[j]); }
        }
        // LDS byte offsets of this lane's pixel fragments (k-step 0) for one tap: staged row = brow + tap offset
        int xad[CR];
        auto tap_addr = [&](int tapoff, int ctap) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                // (opaque: the offsets of all nine taps are loop-invariant, and hipcc would hoist them -- and their four k-step
                // variants -- out of the chunk and tile loops and then spill them)
                int b = brow[j] < 0 ? 0 : brow[j];
                asm volatile("" : "+v"(b));
                const int row = b + tapoff;
                const int c = (pt0 + j) * 32 + r + ctap;       // yq * TW + xq of the pixel this tap reads (the loader's key)
                xad[j] = (row << 7) | ((h ^ ((c >> 1) & 7)) << 4);
            }
        };
        auto read_a = [&](uint4 (&fa)[TN], int wsl, int q) __attribute__((always_inline)) {
            if (LC_DBG & 128) return;
            const char *pw = lds + wsl * WSLOT + rdA + (swa0 ^ (q << 5));
#pragma unroll
            for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4 *>(pw + i * 32 * 128);
        };
        auto read_b = [&](int j, int xsl, int q) __attribute__((always_inline)) {
            if (LC_DBG & 128) return;
            fb[j] = *reinterpret_cast<const uint4 *>(lds + NSW * WSLOT + xsl * XSLOT + (xad[j] ^ (q << 5)));
        };
        // one k-step: the next step's weight fragments first, then per position group its MFMAs and, right behind them, the
        // group's next pixel fragment (sched_barrier: hipcc's scheduler would otherwise sink every read down to its use)
        auto kstep = [&](bool more, int wsl_n, int xsl_n, int q_n) __attribute__((always_inline)) {
            if constexpr (C > 0) {
                if (more) read_a(fan, wsl_n, q_n);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    if (!(LC_DBG & 1)) {
#pragma unroll
                        for (int i = 0; i < TN; ++i) Mma<T>::run(fac[i], fb[j], acc[i][j]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) read_b(j, xsl_n, q_n);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) fac[i] = fan[i];
            }
        };
        int wsl = 0, xsl = 0;                                  // ring slots of the current tap / chunk, running across tiles
        int gidx = gidx0;
        int gtap = 0;                                          // taps done (= barriers passed - 1), running across tiles
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        };
        zero_acc();                                            // (before the barrier: the first fills are still landing)
        LC_BARRIER(0);                                         // barrier 0: the first fills have landed
        for (;;) {
            tap_addr(0, 0);
            if constexpr (C > 0) {
                read_a(fac, wsl, 0);
#pragma unroll
                for (int j = 0; j < C; ++j) read_b(j, xsl, 0);
            }
            for (int cc = 0; cc < cchunks; ++cc) {
                const bool last_chunk = cc + 1 == cchunks;
                auto tap = [&](auto T9) __attribute__((always_inline)) {
                    constexpr int t9 = decltype(T9)::value;
                    const int wsn = wsl + 1 == NSW ? 0 : wsl + 1;
                    // k-steps 0 .. 2 refresh the fragments out of this tap's slots
                    kstep(true, wsl, xsl, 1);
                    kstep(true, wsl, xsl, 2);
                    kstep(true, wsl, xsl, 3);
                    // barrier g + 1: tap g + 1's weights (and after a chunk's last tap the next chunk's pixels) have landed; every
                    // consumer has COMPLETED its last read of this tap's weight slot (and of the chunk's pixel slot): the loaders
                    // refill those slots right behind the barrier, and s_barrier alone orders nothing against LDS reads that
                    // are still queued (gfx950 barriers do not wait for lgkmcnt) -- the explicit wait makes the slot hand-over a
                    // dependency instead of a race that the L2 latency happened to win.  The fragments are
                    // needed by the first MFMA behind the barrier anyway.
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    LC_BARRIER(gtap + 1);
                    ++gtap;
                    __builtin_amdgcn_sched_barrier(0);
                    // k-step 3 refreshes them out of the next tap's
                    if constexpr (t9 < 8) {
                        constexpr int ki = (t9 + 1) / 3, kj = (t9 + 1) % 3;
                        tap_addr(ki * PW + kj, ki * a.TW + kj);
                        kstep(true, wsn, xsl, 0);
                    } else {
                        tap_addr(0, 0);
                        kstep(!last_chunk, wsn, xsl ^ 1, 0);
                    }
                    wsl = wsn;
                };
                tap(std::integral_constant<int, 0>());
                tap(std::integral_constant<int, 1>());
                tap(std::integral_constant<int, 2>());
                tap(std::integral_constant<int, 3>());
                tap(std::integral_constant<int, 4>());
                tap(std::integral_constant<int, 5>());
                tap(std::integral_constant<int, 6>());
                tap(std::integral_constant<int, 7>());
                tap(std::integral_constant<int, 8>());
                xsl ^= 1;
            }
            store_tile(gidx);
            gidx += wpx;
            if (gidx >= chunk_hi) break;
            zero_acc();
        }
    };
    switch (cnt) {
    case 0: main_loop(std::integral_constant<int, 0>()); break;
    case 1: main_loop(std::integral_constant<int, 1>()); break;
    case 2: main_loop(std::integral_constant<int, (CMAX >= 2 ? 2 : CMAX)>()); break;
    case 3: main_loop(std::integral_constant<int, (CMAX >= 3 ? 3 : CMAX)>()); break;
    case 4: main_loop(std::integral_constant<int, (CMAX >= 4 ? 4 : CMAX)>()); break;
    default: main_loop(std::integral_constant<int, CMAX>()); break;
    }
}

// Tile shape of a launch: kind 0 = 128 channels (consumers 2 x 2, each 64 channels x up to 5 position groups, 440 staged
// pixels per chunk), kind 1 = 64 channels (each consumer 32 channels x up to 5 groups, 504 staged pixels).  TH x TW is picked
// per layer: an even TW, (TH + 2)(TW + 2) staged pixels within the slot, TH * TW <= 320 outputs.
struct LcPlan { int kind, TH, TW; };
struct LcKind { int BN, TN, WM, CMAX, xrows; };
static const LcKind LC_KINDS[2] = {{128, 2, 2, 5, 440}, {64, 1, 2, 5, 504}};

static LcPlan lc_plan(int B, int H, int W, int Cn, int Ck)
{
    static DcfOpt ek_o("LC_KIND"), eh_o("LC_TH"), ew_o("LC_TW");
    const char *ek = ek_o.str(), *eh = eh_o.str(), *ew = ew_o.str();
    const int ncu = 256;
    LcPlan best = {-1, 0, 0};
    double best_t = 1e30;
    const int cchunks = Ck / 64;
    for (int kind = 0; kind < 2; ++kind) {
        const LcKind &k = LC_KINDS[kind];
        if (Cn % k.BN) continue;
        if (ek && atoi(ek) != kind) continue;
        const int maxpos = k.WM * k.CMAX * 32;
        for (int TW = 14; TW <= std::min(std::max(W + (W & 1), 14), 254); TW += 2) {     // (TW >= 14: see setup_x)
            if (ew && atoi(ew) != TW) continue;
            for (int TH = 1; TH <= H; ++TH) {
                if (TH * TW > maxpos || (TH + 2) * (TW + 2) > k.xrows) break;
                if (eh && atoi(eh) != TH) continue;
                const int64_t tiles = (int64_t)B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * (Cn / k.BN);
                const int64_t rounds = (tiles + ncu - 1) / ncu;
                const int groups = (TH * TW + 31) / 32;
                const int per_wave = (groups + k.WM - 1) / k.WM;
                // cycles per tap on a CU: the busiest consumer's MFMAs; the loaders' pieces (two waves of each kind, ~60 cycles of
                // a wave per piece); a barrier.  Per tile: the epilogue (stores per consumer) and its share of the first fills
                const double mfma = 4.0 * k.TN * per_wave * 32;
                const double wld = (k.BN / 8) / 2.0 * 60.0;
                const double xld = ((TH + 2) * (TW + 2) / 8.0) / 2.0 / 9.0 * 60.0;
                // (measured on the 128-channel stage: 1670 cycles per tap for 1280 of MFMAs, 1290 for 1024, 1050 for 768; ~6 000
                // per tile for the epilogue, which no MFMA overlaps; ~8 000 until the first tile's first fill has landed)
                const double step = std::max(std::max(mfma, wld), xld) + 300.0;
                const double t = rounds * (9.0 * cchunks * step + 3500.0 + 300.0 * k.TN * per_wave) + 6000.0;
                if (t < best_t) { best_t = t; best = {kind, TH, TW}; }
            }
        }
    }
    return best;
}

}  // namespace

// Called by dcf_conv2d_fwd / dcf_conv2d_dgrad (conv.hip).  Returns DCF_EUNSUPPORTED when the shape is not this kernel's.
// force = 0: only where this kernel measured faster than conv_rs.hip (tools/lc_bench.py, profiles/r04d_lc_bench_*.txt): launches
// of more than one round of workgroups -- a single round is mostly ramp (the first fill, the store burst at the end: ~10 us of a
// ~27 us launch on the 128-channel stage at batch 2), which the persistent loop hides from the second tile on.
int dcf_conv3x3_lc_launch(int dtype, const void *x, const void *w, const float *shift, const void *res, const void *mask, void *y,
                          int B, int H, int W, int Ck, int Cn, int relu, int flip, const char *name_base, double flops, int force, hipStream_t s)
{
    if ((dtype != DCF_BF16 && dtype != DCF_F16) || Ck % 64 || Cn % 64 || Ck < 64 || Cn < 64) return DCF_EUNSUPPORTED;
    if ((int64_t)B * H * W * Ck * 2 >= (1ll << 31) || (int64_t)B * H * W * Cn * 2 >= (1ll << 31) || H >= 65536 || W >= 65536) return DCF_EUNSUPPORTED;
    const LcPlan p = lc_plan(B, H, W, Cn, Ck);
    if (p.kind < 0) return DCF_EUNSUPPORTED;
    LcArgs a;
    a.x = (const char *)x; a.w = (const char *)w; a.shift = shift; a.res = (const char *)res; a.mask = (const char *)mask; a.y = (char *)y;
    a.B = B; a.H = H; a.W = W; a.Ck = Ck; a.Cn = Cn; a.relu = relu; a.flip = flip;
    a.TH = p.TH; a.TW = p.TW;
    a.nty = (H + p.TH - 1) / p.TH; a.ntx = (W + p.TW - 1) / p.TW;
    a.mtiles = B * a.nty * a.ntx;
    a.xbytes = (unsigned)((int64_t)B * H * W * Ck * 2);
    a.wbytes = (unsigned)((int64_t)Cn * 9 * Ck * 2);
    a.ybytes = (unsigned)((int64_t)B * H * W * Cn * 2);
    const int BN = LC_KINDS[p.kind].BN;
    if (!force && (int64_t)a.mtiles * (Cn / BN) < 320) return DCF_EUNSUPPORTED;
    int64_t nwg = (((int64_t)a.mtiles * (Cn / BN) + 7) / 8) * 8;
    nwg = std::min<int64_t>(nwg, 256);                       // persistent workgroups: at most one per CU
    const dim3 grid((unsigned)nwg);
    char name[96];
    snprintf(name, sizeof(name), "%s<lc%d,%dx%d>", name_base, p.kind, p.TH, p.TW);
    const double bytes = (double)a.xbytes + (double)a.wbytes + (double)B * H * W * Cn * 2.0 * (1 + (res ? 1 : 0) + (mask ? 1 : 0));
#define DCF_LC(T_)                                                                                                               \
    do {                                                                                                                         \
        if (p.kind == 0) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_lc<T_, 2, 5, 2, 2, 3, 440>), grid, dim3(512), 0, s, a)); \
        else DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_lc<T_, 1, 5, 2, 2, 4, 504>), grid, dim3(512), 0, s, a)); \
    } while (0)
#ifdef LC_BF16_ONLY            /* tools/rw_variants.sh: half the compile time */
    if (dtype == DCF_F16) return DCF_EUNSUPPORTED;
    DCF_LC(bf16_t);
#else
    if (dtype == DCF_F16) DCF_LC(f16_t); else DCF_LC(bf16_t);
#endif
#undef DCF_LC
    return DCF_OK;
}

#ifdef LC_STAMP
extern "C" int dcf_lc_stamps_read(long long *dst, int *dims)
{
    dims[0] = LC_STAMP_WGS; dims[1] = 8; dims[2] = LC_STAMP_MAXG; dims[3] = 2;
    DCF_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_lc_stamps), sizeof(long long) * LC_STAMP_WGS * 8 * LC_STAMP_MAXG * 2));
    return DCF_OK;
}
extern "C" int dcf_lc_stamps_clear(void)
{
    static long long zeros[LC_STAMP_WGS * 8 * LC_STAMP_MAXG * 2];
    DCF_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_lc_stamps), zeros, sizeof(zeros)));
    return DCF_OK;
}
#endif
