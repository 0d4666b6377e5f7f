// conv_lc.hip -- row-sharing implicit GEMM with LOADER and CONSUMER waves, for the 3x3 / stride-1 / pad-1 convolutions
// (forward and input gradient), 16-bit element types, gfx950.
//
// The layers are those of conv_rs.hip (/root/reference/model.py:15-28 ResidualBlock bodies, :153 conv3; the camera trunk's
// BasicBlocks); the staging is conv_rs.hip's too: [Cn][9][Ck] weights and row-shared pixel tiles (one staged tile of BM + 2
// padded positions per kernel row x 64-channel chunk serves the three horizontal taps at row offsets 0 / 1 / 2) go through
// LDS-DMA rings.  What changes is WHO issues what.  Measured on conv_rs.hip / conv_rw.hip (profiles/r04a_sq_*.csv,
// profiles/r04b_rw_ablation.txt): the MFMA + fragment-read stream alone runs at the matrix pipes' pace, but every
// vector-memory instruction a wave issues (LDS-DMA piece or load: ~23 cycles of the CU's address path each, 60-180 cycles of
// the issuing wave's time) is time in which that wave issues no MFMA, and with every wave doing both jobs behind one barrier
// per tap the waves were parked 60 % of their cycles.  Here:
//
//   * waves 0-3 (one per SIMD) are CONSUMERS: fragment reads one k-step ahead + MFMAs, no vector-memory instruction in the
//     loop, a 2 x 2 arrangement of (TN x 32 channels) x (C x 32 positions) register tiles;
//   * waves 4-5 are WEIGHT LOADERS, waves 6-7 PIXEL LOADERS (one per SIMD, beside a consumer): they issue every LDS-DMA piece.
//     Two kinds of loader because s_waitcnt vmcnt retires in issue order: a wave that loaded both would wait for its pixel
//     pieces (HBM, a stage or two ahead) whenever it waits for a tap's weights (L2, two taps ahead);
//   * rings: NSW weight slots (one tap each), NSX pixel slots (one stage = kernel row x chunk each), both running across tile
//     boundaries of the persistent workgroup -- the loaders are filling the next tile's first slots while the consumers store
//     the current tile, whose stores then drain under the next tile's MFMAs (nothing in a consumer's loop waits on vmcnt);
//   * one s_barrier per tap, joined by all eight waves, placed between k-steps 2 and 3 of the tap: barrier g + 1 (inside tap g)
//     tells the consumers that tap g + 1's weights (and, at a stage's end, the next stage's pixel tile) have landed -- the
//     loaders wait for their own pieces with a counted vmcnt before joining -- and tells the loaders that every consumer has
//     issued its last read of tap g's weight slot (and of the stage's pixel slot), which they refill right after.
//
// Same K order (kernel row, chunk, tap, k-step) and fragments as conv_rs.hip: bit-identical results.
// dgrad = the same kernel on the [Cin][tap][Cout] weight image with the taps mirrored.
// Algorithmic work per launch: 2*B*H*W*Cout*Cin*9 flop; bytes B*H*W*(Cin + Cout)*2 + weights (+ residual / mask reads).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "dcf_common.h"
#include "conv_common.h"

// Timing ablations, COMPILE-TIME only (tools/lc_variants.sh builds one library per mask): 1 no MFMAs, 2 pixel DMA reads nothing,
// 4 no epilogue, 16 weight DMA reads nothing, 32 pixel DMA not issued, 64 weight DMA not issued, 128 no LDS fragment reads,
// 256 the loaders do not wait for their pieces, 512 only the middle kernel row's pixel tiles are fetched (a third of the pixel traffic).  Results are wrong in those builds; shipped with 0.
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
        if (st__) g_lc_stamps[blockIdx.x][threadIdx.x >> 6][g__][0] = __builtin_amdgcn_s_memtime();                         \
        __builtin_amdgcn_s_barrier();                                                                                       \
        if (st__) g_lc_stamps[blockIdx.x][threadIdx.x >> 6][g__][1] = __builtin_amdgcn_s_memtime();                         \
    } while (0)
#else
#define LC_BARRIER(gg) __builtin_amdgcn_s_barrier()
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
    int npt;              // 32-position tiles per workgroup
    int mtiles;           // position tiles of the launch
    int Q;                // padded positions B*H*(W+2)
    unsigned xbytes, wbytes, ybytes;
};

typedef unsigned lc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lc_gst16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, uint4 data)
{
    const lc_u32x4 d = {data.x, data.y, data.z, data.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)voff, 0, 0);
}
__device__ __forceinline__ void lc_keep(const f32x16 &v) { asm volatile("" ::"v"(v)); }
__device__ __forceinline__ void lc_opaque(uint4 &v) { asm volatile("" : "+v"(v)); }
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
// the workgroup's npt position tiles (at most CMAX).  NSW weight slots (taps), NSX pixel slots (stages), BMMAX = most
// positions of a workgroup tile (sizes the pixel slot).
template <typename T, int TN, int CMAX, int WN, int WM, int NSW, int NSX, int BMMAX>
__global__ void __launch_bounds__(512) k_conv3x3_lc(LcArgs a)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    static_assert(WN * WM == 4, "one consumer per SIMD");
    static_assert(NSW >= 3 && NSX >= 3, "a slot is refilled while the next one is read and the one after is certified");
    static_assert(WM * CMAX * 32 >= BMMAX && BMMAX % 32 == 0, "position tiles");
    constexpr int BN = WN * TN * 32;
    constexpr int WSLOT = BN * 128;
    constexpr int XROWS = (BMMAX + 2 + 7) / 8 * 8;
    constexpr int XSLOT = XROWS * 128;
    constexpr int NWL = BN / 8 / 2;                           // weight pieces (8 rows x 128 B) per weight loader and tap
    constexpr int PXL = (XROWS / 8 + 1) / 2;                  // most pixel pieces per pixel loader and stage
    constexpr int PX3 = (PXL + 2) / 3;                        // pixel pieces per pixel loader and tap (a third of a stage's)
    static_assert(NWL >= 1 && (NSW - 2) * NWL < 64 && (NSX - 2) * 3 * PX3 < 64, "vmcnt range");
    // + 1 KiB that the pixel loaders' surplus pieces are written to (see issue_chunk)
    static_assert(NSW * WSLOT + NSX * XSLOT + 1024 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char lds[NSW * WSLOT + NSX * XSLOT + 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    // PERSISTENT workgroups, XCD-aware tile order (speed only): XCD x = blockIdx & 7 owns the x-th contiguous chunk of the
    // (position tile, channel tile) list, channel tiles fastest; its workgroups take the chunk's tiles round-robin.
    const int nt = a.Cn / BN;
    const int nblk = a.mtiles * nt;
    const int chunk = (nblk + 7) >> 3;
    const int wpx = gridDim.x >> 3;                   // workgroups per XCD
    const int chunk_lo = (blockIdx.x & 7) * chunk, chunk_hi = min(chunk_lo + chunk, nblk);
    const int gidx0 = chunk_lo + (blockIdx.x >> 3);
    if (gidx0 >= chunk_hi) return;
    const int ntile_wg = (chunk_hi - gidx0 + wpx - 1) / wpx;
    const int BM = a.npt * 32;
    const int Wp = a.W + 2, BH = a.B * a.H;
    const int rowbytes = a.Ck * 2;
    const int cchunks = rowbytes / 128;
    const int nstage = 3 * cchunks, ntaps = 3 * nstage;
    const int G = ntile_wg * ntaps;                   // taps of this workgroup = barriers after the first
    const int rowpitch = a.W * rowbytes;
    constexpr unsigned OOB = 0xFFFFFF00u;
    const unsigned ldsW0 = lds_addr(lds), ldsX0 = ldsW0 + NSW * WSLOT;
    const int l8 = lane >> 3, lc = lane & 7;          // DMA lane = (row of the 8-row piece, 16-byte chunk position)

    if (wid >= 6) {
        // ================================================================ PIXEL LOADER (lx = 0, 1: pieces lx, lx + 2, ...)
        // LDS row i of a slot = padded position q0 - 1 + i; position lc of row R holds source chunk lc ^ ((R >> 1) & 7): the 16
        // rows of a ds_read_b128 lane group sit on distinct banks (also at row offsets 1 and 2, i.e. for all three taps).
        const int lx = wid - 6;
        const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
        const int npieces = (BM + 2 + 7) >> 3;
        const int cntx = __builtin_amdgcn_readfirstlane(lx < npieces ? (npieces - 1 - lx) / 2 + 1 : 0);
        int xbase[PXL], xok[PXL];
        auto setup_x = [&](int gi) __attribute__((always_inline)) {
            const int q0 = (gi / nt) * BM;                     // first padded position of the tile
            // ONE division per tile: this wave sets up every other piece of the tile (up to PXL of them), and a quotient +
            // remainder pair per piece (~100 instructions each) stalled the whole workgroup for microseconds at every tile
            // boundary; a lane's next piece is 16 positions further on
            int p = q0 - 1 + lx * 8 + l8;
            int R = p >= 0 ? p / Wp : 0;
            int c = p - R * Wp;
            int oh = R % a.H;
#pragma unroll
            for (int j = 0; j < PXL; ++j) {
                const int i = (lx + 2 * j) * 8 + l8;
                const bool live = (i < BM + 2) && (p >= 0) && (R < BH) && (c >= 1) && (c <= a.W);
                xbase[j] = (R * a.W + c - 1) * rowbytes + ((lc ^ ((i >> 1) & 7)) * 16);
                xok[j] = live ? ((oh >= 1 ? 1 : 0) | 2 | (oh + 1 < a.H ? 4 : 0)) : 0;
                p += 16; c += 16;
                while (c >= Wp) { c -= Wp; ++R; if (++oh == a.H) oh = 0; }
            }
        };
        int gi = gidx0, ki = 0, cc = 0;                        // fill cursor: tile, kernel row, chunk
        bool live = true;
        setup_x(gi);
        // Pieces j = c3 (mod 3) of the cursor's stage.  STRAIGHT-LINE code, always PX3 pieces: a first version that branched around
        // the pieces this wave does not have (j >= cntx, j % 3 != c3 with c3 a run-time value) spent ~190 cycles per piece in
        // taken branches and instruction fetches -- the whole workgroup waited for its pixel loaders at every barrier.  A piece
        // the wave does not have goes out of range (zeros) into a spare KiB of LDS; every fill is then 3 PX3 pieces and the
        // certifying wait is an immediate.
        const unsigned ldsDump = ldsX0 + NSX * XSLOT;
        auto issue_chunk = [&](auto C3, int slot) __attribute__((always_inline)) {
            constexpr int c3 = decltype(C3)::value;
            if (LC_DBG & 32) return;
            const int xst = (ki - 1) * rowpitch + cc * 128;
#pragma unroll
            for (int k = 0; k < PX3; ++k) {
                const int j = 3 * k + c3;
                if (j < PXL) {
                    const bool real = j < cntx;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(real ? ldsX0 + slot * XSLOT + (lx + 2 * j) * 1024 : ldsDump);
                    glds16(srcX, (real && ((xok[j] >> ki) & 1) && !(LC_DBG & 2) && (!(LC_DBG & 512) || ki == 1)) ? (unsigned)(xbase[j] + xst) : OOB, dst);
                } else {
                    glds16(srcX, OOB, ldsDump);
                }
            }
        };
        auto issue_third = [&](int c3, int slot) __attribute__((always_inline)) {
            if (c3 == 0) issue_chunk(std::integral_constant<int, 0>(), slot);
            else if (c3 == 1) issue_chunk(std::integral_constant<int, 1>(), slot);
            else issue_chunk(std::integral_constant<int, 2>(), slot);
        };
        auto advance = [&]() __attribute__((always_inline)) {
            if (++cc == cchunks) {
                cc = 0;
                if (++ki == 3) {
                    ki = 0;
                    gi += wpx;
                    live = gi < chunk_hi;
                    if (live) setup_x(gi);
                }
            }
        };
        const int V = ntile_wg * nstage;                       // stages (fills) of this workgroup
        for (int v = 0; v < NSX; ++v)
            if (live) {
                issue_chunk(std::integral_constant<int, 0>(), v);
                issue_chunk(std::integral_constant<int, 1>(), v);
                issue_chunk(std::integral_constant<int, 2>(), v);
                advance();
            }
        wait_vmcnt<0>();
        LC_BARRIER(0);                                         // barrier 0
        // Fill v >= NSX goes into the slot of stage v - NSX, free once barrier 3 (v - NSX) + 3 has passed: its three thirds are
        // issued behind barriers 3 (v - NSX) + 3, + 4, + 5.  Barrier 3 vq certifies stage vq: after its last piece this wave has
        // issued the whole fills vq + 1 .. vq + NSX - 2 (3 PX3 pieces each) and nothing else.
        int slot = 0;                                          // slot of the fill being issued (= its stage index mod NSX)
        for (int g = 0; g < G; ++g) {
            const int g1 = g + 1;
            if (g1 % 3 == 0) {
                const int vq = g1 / 3;
                if (vq < V && !(LC_DBG & 256)) {
                    if (vq + NSX - 2 < V) wait_vmcnt<(NSX - 2) * 3 * PX3>(); else wait_vmcnt<0>();
                }
            }
            LC_BARRIER(g + 1);                                 // barrier g + 1
            if (g >= 2 && live) {
                const int c3 = (g - 2) % 3;
                issue_third(c3, slot);
                if (c3 == 2) { advance(); slot = slot + 1 == NSX ? 0 : slot + 1; }
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
        int gi = gidx0, ki = 0, cc = 0, kj = 0, slot = 0;      // fill cursor: tile, kernel row, chunk, tap; ring slot
        bool live = true;
        setup_w(gi);
        auto issue_w = [&]() __attribute__((always_inline)) {
            if (!live) return;
            const int tapidx = a.flip ? 8 - (3 * ki + kj) : 3 * ki + kj;
            const unsigned koff = (unsigned)(tapidx * rowbytes + cc * 128);
            const unsigned dst = __builtin_amdgcn_readfirstlane(ldsW0 + slot * WSLOT + lw * 1024);
#pragma unroll
            for (int j = 0; j < NWL; ++j)
                if (!(LC_DBG & 64)) glds16(srcW, (LC_DBG & 16) ? OOB : wbase[j] + koff, dst + 2 * j * 1024);
            slot = slot + 1 == NSW ? 0 : slot + 1;
            if (++kj == 3) {
                kj = 0;
                if (++cc == cchunks) {
                    cc = 0;
                    if (++ki == 3) {
                        ki = 0;
                        gi += wpx;
                        live = gi < chunk_hi;
                        if (live) setup_w(gi);
                    }
                }
            }
        };
        for (int u = 0; u < NSW; ++u) issue_w();               // fills 0 .. NSW-1
        wait_vmcnt<0>();
        LC_BARRIER(0);                                         // barrier 0
        // Barrier g + 1 certifies tap g + 1 (fill g + 1): behind it this wave has issued fills g + 2 .. g + NSW - 1.  After the
        // barrier tap g's slot is free: fill g + NSW.
        for (int g = 0; g < G; ++g) {
            if (!(LC_DBG & 256)) { if (g + NSW - 1 < G) wait_vmcnt<(NSW - 2) * NWL>(); else wait_vmcnt<0>(); }
            LC_BARRIER(g + 1);                                 // barrier g + 1
            issue_w();
        }
        wait_vmcnt<0>();
        return;
    }

    // ==================================================================== CONSUMERS
    const int wn = wid / WM, wm = wid % WM;
    const int base = a.npt / WM, rem = a.npt - base * WM;
    const int cnt = base + (wm < rem ? 1 : 0);
    const int pt0 = wm * base + min(wm, rem);
    const __amdgpu_buffer_rsrc_t dstY = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.ybytes, 0x00020000);

    // fragment reads: k-step q, lane half h reads source chunk 2 q + h of its row, stored at position (2 q + h) ^ key(row):
    // byte offset ((h ^ key) << 4) ^ (q << 5), key = (row >> 1) & 7 (weights: row r; pixels: row r + kj, + multiples of 32)
    const int swa0 = (h ^ ((r >> 1) & 7)) << 4;
    const int rdA = (wn * TN * 32 + r) * 128;
    const int rdX = NSW * WSLOT + (pt0 * 32 + r) * 128;

    f32x16 acc[TN][CMAX];
    const T *res = reinterpret_cast<const T *>(a.res);
    const T *mask = reinterpret_cast<const T *>(a.mask);
    // epilogue of one tile (as k_conv3x3_rs): v = acc + shift + res ; relu ; v *= (mask > 0) ; 8 consecutive channels per access;
    // one channel tile at a time, so that the residual / mask vectors in flight stay at 2 CMAX registers x 4
    auto store_tile = [&](int q0c, int n0c) __attribute__((always_inline)) {
        if (LC_DBG & 4) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < CMAX; ++j) lc_keep(acc[i][j]);
            return;
        }
        int mrow[CMAX];
        bool valid[CMAX];
#pragma unroll
        for (int j = 0; j < CMAX; ++j) {
            const int p = q0c + (pt0 + j) * 32 + r;
            const int R = p / Wp, c = p - R * Wp;
            valid[j] = (j < cnt) && !(p >= a.Q || c < 1 || c > a.W);      // padding position: no output
            mrow[j] = R * a.W + c - 1;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            auto voff = [&](int j, int pp) { return (size_t)mrow[j] * a.Cn + n0c + (wn * TN + i) * 32 + 16 * pp + 8 * h; };
#pragma unroll
            for (int j = 0; j < CMAX; ++j) acc_rows8(acc[i][j]);
            if (res) {
                uint4 rr[CMAX][2];
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        rr[j][pp] = valid[j] ? *reinterpret_cast<const uint4 *>(res + voff(j, pp)) : make_uint4(0, 0, 0, 0);
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
                    for (int pp = 0; pp < 2; ++pp)
                        mm[j][pp] = valid[j] ? *reinterpret_cast<const uint4 *>(mask + voff(j, pp)) : make_uint4(0, 0, 0, 0);
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
                    lc_gst16(dstY, valid[j] ? (unsigned)(voff(j, pp) * sizeof(T)) : OOB, lc_pack8<T>(v));
                }
            }
        }
    };

    // Main loop, specialised on the wave's tile count C (wave-uniform).
    auto main_loop = [&](auto CNT) {
        constexpr int C = decltype(CNT)::value;
        constexpr int CR = C > 0 ? C : 1;
        uint4 fac[TN], fbc[CR], fan[TN], fbn[CR];
        if (LC_DBG & 128) {
#pragma unroll
            for (int i = 0; i < TN; ++i) { fac[i] = make_uint4(lane, 1, 2, 3); lc_opaque(fac[i]); fan[i] = fac[i]; }
#pragma unroll
            for (int j = 0; j < CR; ++j) { fbc[j] = make_uint4(lane, 5, 6, 7); lc_opaque(fbc[j]); fbn[j] = fbc[j]; }
        }
        // fragments of k-step q of tap kj: weights out of slot wsl, pixels out of slot xsl
        auto read_f = [&](uint4 (&fa)[TN], uint4 (&fb)[CR], int wsl, int xsl, int kj, int q) __attribute__((always_inline)) {
            if (LC_DBG & 128) return;
            if constexpr (C > 0) {
                const char *pw = lds + wsl * WSLOT + rdA + (swa0 ^ (q << 5));
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4 *>(pw + i * 32 * 128);
                const char *px = lds + xsl * XSLOT + rdX + kj * 128 + (((h ^ (((r + kj) >> 1) & 7)) << 4) ^ (q << 5));
#pragma unroll
                for (int j = 0; j < C; ++j) fb[j] = *reinterpret_cast<const uint4 *>(px + j * 32 * 128);
            }
        };
        auto mma = [&]() __attribute__((always_inline)) {
            if constexpr (C > 0) {
                if (!(LC_DBG & 1)) {
#pragma unroll
                    for (int j = 0; j < C; ++j)
#pragma unroll
                        for (int i = 0; i < TN; ++i) Mma<T>::run(fac[i], fbc[j], acc[i][j]);
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) fac[i] = fan[i];
#pragma unroll
                for (int j = 0; j < C; ++j) fbc[j] = fbn[j];
            }
        };
        int wsl = 0, xsl = 0;                                  // ring slots of the current tap / stage, running across tiles
        int gidx = gidx0;
        int gtap = 0;                                          // taps done (= barriers passed - 1), running across tiles
        LC_BARRIER(0);                                         // barrier 0: the first fills have landed
        for (;;) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < CMAX; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
            read_f(fac, fbc, wsl, xsl, 0, 0);
            for (int s = 0; s < nstage; ++s) {
                const int xsn = xsl + 1 == NSX ? 0 : xsl + 1;
                const bool last_stage = s + 1 == nstage;
                auto tap = [&](auto KJ) __attribute__((always_inline)) {
                    constexpr int kj = decltype(KJ)::value;
                    const int wsn = wsl + 1 == NSW ? 0 : wsl + 1;
                    // k-steps 0 .. 2: the next k-step's fragments first (hipcc's scheduler would sink the reads to their uses:
                    // sched_barrier keeps them a whole MFMA group ahead), then this one's MFMAs
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        read_f(fan, fbn, wsl, xsl, kj, q + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        mma();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // barrier g + 1: tap g + 1's weights (and after a stage's last tap the next stage's pixels) have landed;
                    // every consumer has issued its last read of this tap's weight slot (and of the stage's pixel slot)
                    LC_BARRIER(gtap + 1);
                    ++gtap;
                    __builtin_amdgcn_sched_barrier(0);
                    if (kj < 2) read_f(fan, fbn, wsn, xsl, kj + 1, 0);
                    else if (!last_stage) read_f(fan, fbn, wsn, xsn, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma();
                    __builtin_amdgcn_sched_barrier(0);
                    wsl = wsn;
                };
                tap(std::integral_constant<int, 0>());
                tap(std::integral_constant<int, 1>());
                tap(std::integral_constant<int, 2>());
                xsl = xsn;
            }
            const int q0c = (gidx / nt) * BM, n0c = (gidx % nt) * BN;
            store_tile(q0c, n0c);
            gidx += wpx;
            if (gidx >= chunk_hi) break;
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

// Tile shape of a launch: kind 0 = 128 channels x up to 288 positions (consumers 2 x 2, each 64 channels x up to 5 position
// tiles), kind 1 = 64 channels x up to 320 positions (consumers 2 x 2, each 32 channels x up to 5 position tiles).
struct LcPlan { int kind, npt; };
struct LcKind { int BN, TN, WM, CMAX, maxnpt; };
static const LcKind LC_KINDS[2] = {{128, 2, 2, 5, 9}, {64, 1, 2, 5, 10}};

static LcPlan lc_plan(int64_t Q, int Cn)
{
    static DcfOpt ek_o("LC_KIND"), en_o("LC_NPT");
    const char *ek = ek_o.str(), *en = en_o.str();
    const int ncu = 256;
    LcPlan best = {-1, 0};
    double best_t = 1e30;
    for (int kind = 0; kind < 2; ++kind) {
        const LcKind &k = LC_KINDS[kind];
        if (Cn % k.BN) continue;
        if (ek && atoi(ek) != kind) continue;
        for (int npt = 1; npt <= k.maxnpt; ++npt) {
            if (en && atoi(en) != npt) continue;
            const int64_t tiles = (Q + 32 * npt - 1) / (32 * npt) * (Cn / k.BN);
            const int64_t rounds = (tiles + ncu - 1) / ncu;
            const int per_wave = (npt + k.WM - 1) / k.WM;
            // cycles per tap on a CU: the busiest consumer's MFMAs; the loaders' pieces through the CU's address path (~23
            // cycles each); a barrier; per tile the epilogue and the first fills
            const double mfma = 4.0 * k.TN * per_wave * 32;
            const double pieces = (k.BN / 8 + (32.0 * npt + 2) / 8.0 / 3.0) * 23.0;
            const double step = std::max(mfma, pieces) + 60.0;
            const double t = rounds * (step + 25.0 * k.TN * per_wave /* epilogue share per tap */);
            if (t < best_t) { best_t = t; best = {kind, npt}; }
        }
    }
    return best;
}

}  // namespace

// Called by dcf_conv2d_fwd / dcf_conv2d_dgrad (conv.hip).  Returns DCF_EUNSUPPORTED when the shape is not this kernel's.
int dcf_conv3x3_lc_launch(int dtype, const void *x, const void *w, const float *shift, const void *res, const void *mask, void *y,
                          int B, int H, int W, int Ck, int Cn, int relu, int flip, const char *name_base, double flops, hipStream_t s)
{
    if ((dtype != DCF_BF16 && dtype != DCF_F16) || Ck % 64 || Cn % 64 || Ck < 64 || Cn < 64) return DCF_EUNSUPPORTED;
    const int64_t Q = (int64_t)B * H * (W + 2);
    if (Q >= (1ll << 30) || (int64_t)B * H * W * Ck * 2 >= (1ll << 31) || (int64_t)B * H * W * Cn * 2 >= 0xFFFFFF00ll) return DCF_EUNSUPPORTED;
    const LcPlan p = lc_plan(Q, Cn);
    if (p.kind < 0) return DCF_EUNSUPPORTED;
    LcArgs a;
    a.x = (const char *)x; a.w = (const char *)w; a.shift = shift; a.res = (const char *)res; a.mask = (const char *)mask; a.y = (char *)y;
    a.B = B; a.H = H; a.W = W; a.Ck = Ck; a.Cn = Cn; a.relu = relu; a.flip = flip;
    a.npt = p.npt; a.Q = (int)Q;
    a.mtiles = (int)((Q + 32 * p.npt - 1) / (32 * p.npt));
    a.xbytes = (unsigned)((int64_t)B * H * W * Ck * 2);
    a.wbytes = (unsigned)((int64_t)Cn * 9 * Ck * 2);
    a.ybytes = (unsigned)((int64_t)B * H * W * Cn * 2);
    const int BN = LC_KINDS[p.kind].BN;
    int64_t nwg = (((int64_t)a.mtiles * (Cn / BN) + 7) / 8) * 8;
    nwg = std::min<int64_t>(nwg, 256);                       // persistent workgroups: at most one per CU
    const dim3 grid((unsigned)nwg);
    char name[96];
    snprintf(name, sizeof(name), "%s<lc%d,%d>", name_base, p.kind, p.npt);
    const double bytes = (double)a.xbytes + (double)a.wbytes + (double)B * H * W * Cn * 2.0 * (1 + (res ? 1 : 0) + (mask ? 1 : 0));
#define DCF_LC(T_)                                                                                                               \
    do {                                                                                                                         \
        if (p.kind == 0) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_lc<T_, 2, 5, 2, 2, 3, 3, 288>), grid, dim3(512), 0, s, a)); \
        else DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_lc<T_, 1, 5, 2, 2, 4, 3, 320>), grid, dim3(512), 0, s, a)); \
    } while (0)
#ifdef LC_BF16_ONLY            /* tools/lc_variants.sh: half the compile time */
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
