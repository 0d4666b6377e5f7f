// conv_wgs.hip -- weight gradient of the 3x3 / stride-1 / pad-1 convolutions with SHARED staging (16-bit types, gfx950).
//
// The row-sharing LDS-DMA kernel of conv.hip (k_conv_wgrad3g) gives every wave its own pixel range and its own ring: a wave
// stages a 32-pixel gy tile (4 KB) and a 34-pixel x tile (4.25 KB) for 24 MFMAs -- 93 flop per staged byte, and with the
// ~21 B/clk a CU pulls out of its L2 into LDS its loop is half idle (0.28-0.31 of the MFMA peak).  Here the waves of a
// workgroup are the 64 x 64 QUADRANTS of one (64 A) x (64 BC) output tile (instantiated: A = BC = 2, 128 x 128 channels)
// walking the SAME pixels: a stage of A gy sub-tiles + BC x sub-tiles (16.5 KB for 2 x 2) feeds
// 24 A BC MFMAs -- 190 resp. 139 flop per staged byte.  Two such groups per workgroup take the two halves of the
// workgroup's pixel range (two waves per SIMD hide each other's waits) and meet in LDS at the end, so a workgroup still
// writes ONE fp32 slab tile; with the layers of a backward pass issued in grouped launches a layer does not have to fill
// the chip on its own, so the number of pixel ranges (= slabs) per layer stays where it was.
//
//   * same padded-image walk as k_conv_wgrad3g: positions of rows of W + 2; tap kj = the staged x tile read kj rows down;
//     a workgroup owns one kernel row ki of one (range, co tile, ci tile) unit; XCD-aware unit order.
//   * LDS-DMA ring per group (inline asm `buffer_load ... lds`, counted vmcnt, out-of-range = zeros), one barrier per stage.
//     Every wave issues the same number of 1-KiB pieces per stage (the remainder are out-of-range pieces into a scratch
//     KiB, also past the end of the range), so the waits are immediates.
//   * fixed-order reduction (group 0 + group 1) -> weight gradients stay bitwise reproducible.
// Algorithmic work per layer: 2*B*H*W*Cout*Cin*9 flop; x and gy read once, nsplit fp32 slab sets written.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "dcf_common.h"
#include "conv_common.h"

// timing ablations, compile time only (-DDCF_WGS_DBG_MASK=n): 1 no MFMA, 2 no DMA, 4 no epilogue, 8 no LDS reads, 16 no main loop
#ifndef DCF_WGS_DBG_MASK
#define DCF_WGS_DBG_MASK 0
#endif

namespace {

constexpr int DBG = DCF_WGS_DBG_MASK;

struct WsArgs {
    const char *x;    // [B][H][W][Cin]
    const char *gy;   // [B][H][W][Cout]
    float *slabs;     // [nsplit][Cout][9][Cin]
    float *gsum;      // [4*nsplit][Cout] or null
    int B, H, W, Cin, Cout;
    int M;            // padded positions B*H*(W+2)
    int nsplit, per_split;   // positions per workgroup (multiple of 32 G)
    int co_tiles, ci_tiles;
    unsigned xbytes, gbytes;
    int rot;          // XCD of the layer's first pixel range
    int upx;          // units per XCD (dcf_wgrad_upx, conv.hip)
};

template <typename T, int A, int BC, int G, int NS>
__device__ __forceinline__ void wgs_body(const WsArgs &a, const int bid)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    constexpr int NQ = A * BC, NW = NQ * G;
    constexpr int PK = 32, XROWS = PK + 2;
    constexpr int NA = 4, NB = 5;                       // 1-KiB pieces (8 rows x 128 B) per gy / x sub-tile
    constexpr int NP = A * NA + BC * NB;                // pieces per stage and group
    constexpr int PPW = (NP + NQ - 1) / NQ;             // pieces per wave and stage (padded with out-of-range ones)
    constexpr int SLOT = NP * 1024;
    constexpr int RING = G * NS * SLOT;
    constexpr int RED = (G - 1) * NQ * 16 * 1024;       // one tap's 64 x 64 tile of every wave of the groups >= 1
    constexpr int LDS_BYTES = (RING > RED ? RING : RED) + 1024;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert((NS - 1) * PPW < 60, "vmcnt range");
    static_assert(PPW <= 3 * (PK / 16), "one piece per MFMA group");
    __shared__ __attribute__((aligned(1024))) char lds_all[LDS_BYTES];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wid / NQ, wq = wid - grp * NQ;
    const int qa = wq / BC, qb = wq - qa * BC;          // this wave's quadrant: co sub-tile qa, ci sub-tile qb
    const unsigned lds0 = lds_addr(lds_all);
    const unsigned ring0 = lds0 + grp * NS * SLOT;
    const unsigned dummy = lds0 + (RING > RED ? RING : RED);

    // XCD-aware unit order as in k_conv_wgrad3g: workgroup bid runs on XCD bid & 7, and an XCD owns a contiguous run of
    // (range, co tile, ci tile) units x 3 kernel rows -- the three rows of a unit share x and gy, neighbouring units one of
    // them: L2 hits.  rot = the XCD the layer's first run goes to (the one after the previous layer's last: the few long
    // units of a jointly planned layer must not pile up on the first XCDs).
    const int tiles2 = a.co_tiles * a.ci_tiles;
    // (units per XCD: dcf_wgrad_upx, conv.hip -- dealt evenly by default; option WGRAD_RANGE_XCD=1 = whole pixel ranges per XCD:
    // less traffic, longer launches)
    const int units = tiles2 * a.nsplit, upx = a.upx;
    const int slot_id = bid >> 3;
    const int unit = (((bid & 7) - a.rot) & 7) * upx + slot_id / 3;
    if (unit >= units) return;
    const int ki = slot_id % 3;
    const int slab_id = unit / tiles2;
    const int t2 = unit - slab_id * tiles2;
    const int cit = t2 % a.ci_tiles, cot = t2 / a.ci_tiles;
    const int co0 = cot * A * 64, ci0 = cit * BC * 64;
    const int Wp = a.W + 2, BH = a.B * a.H, Wo = a.W, Ho = a.H;
    const int gspan = a.per_split / G;
    const int q_begin = slab_id * a.per_split + grp * gspan;
    const int q_end = min(min(q_begin + gspan, slab_id * a.per_split + a.per_split), a.M);
    const int nst = (DBG & 16) ? 0 : gspan / PK;         // the same for every group: the barriers line up

    f32x16 acc[3][2][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[k][i][j][q] = 0.f;
    const bool do_sum = (a.gsum != nullptr) && (ki == 0) && (cit == 0) && (qb == 0);
    float fsum[2] = {0.f, 0.f};

    // ---- DMA side.  Lane = (row lr of the 8-row piece, 16-B chunk); 128-B rows, 64-B halves swapped on odd row pairs (the
    // four rows of a transposed read then sit on distinct banks); the swizzle goes on the SOURCE chunk.
    const int rowA = a.Cout * 2, pixB = a.Cin * 2;
    const int lr = lane >> 3;
    const int ch = (lane & 7) ^ (((lr >> 1) & 1) << 2);
    const __amdgpu_buffer_rsrc_t srcG = __builtin_amdgcn_make_buffer_rsrc((void *)a.gy, 0, a.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int lrA = opaque(lr * rowA + ch * 16), lrB = opaque(lr * pixB + ch * 16);
    const int lro = opaque(lr);
    // this wave's pieces: p = wq + k NQ -> (kind, sub-tile, piece of the sub-tile); wave-uniform, loop-invariant
    int pkind[PPW], pcol[PPW], pj[PPW];
    unsigned pdst[PPW];
    bool pch[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int p = wq + k * NQ;
        if (p < A * NA) {
            const int s = p / NA;
            pkind[k] = 0; pj[k] = p - s * NA;
            pcol[k] = (co0 + s * 64) * 2;
            pdst[k] = (unsigned)p * 1024;
            pch[k] = ch * 16 < min(64, a.Cout - (co0 + s * 64)) * 2;
        } else if (p < NP) {
            const int pp = p - A * NA, s = pp / NB;
            pkind[k] = 1; pj[k] = pp - s * NB;
            pcol[k] = (ci0 + s * 64) * 2;
            pdst[k] = (unsigned)p * 1024;
            pch[k] = ch * 16 < min(64, a.Cin - (ci0 + s * 64)) * 2;
        } else {
            pkind[k] = 2; pj[k] = 0; pcol[k] = 0; pdst[k] = 0; pch[k] = false;
        }
    }
    int sq = q_begin;                                    // gy rows of the stage start at position sq, x rows at sq - 1
    int sR = q_begin / Wp, sC = q_begin - sR * Wp;
    int xR, xC, xOh;
    if (q_begin == 0) { xR = -1; xC = Wp - 1; xOh = Ho - 1; }
    else { xR = (q_begin - 1) / Wp; xC = (q_begin - 1) - xR * Wp; xOh = xR % Ho; }
    // one stage's issue = prep (wave-uniform scalars of the stage) + PPW pieces + advance; the main loop spreads the pieces
    // over the stage's MFMA groups: the LDS-DMA path takes ~20 B/clk per CU, and a burst of every wave's pieces right after
    // the barrier holds all the waves at their issue while the matrix pipes idle
    unsigned sbase = 0;
    int rem = 0, baseA = 0, baseB = 0;
    bool ok0 = false, ok1 = false;
    auto prep = [&](int is) {
        sbase = __builtin_amdgcn_readfirstlane(ring0 + is * SLOT);
        rem = q_end - sq;
        baseA = __builtin_amdgcn_readfirstlane((sR * Wo + sC - 1) * rowA);
        const int ih0 = xOh + ki - 1;
        const int oh1 = xOh + 1 == Ho ? 0 : xOh + 1;
        const int ih1 = oh1 + ki - 1;
        ok0 = (xR >= 0) & (xR < BH) & (ih0 >= 0) & (ih0 < a.H);
        ok1 = (xR + 1 < BH) & (ih1 >= 0) & (ih1 < a.H);
        baseB = __builtin_amdgcn_readfirstlane(((xR + ki - 1) * a.W + xC - 1) * pixB);
    };
    auto piece = [&](int k) {
        if constexpr ((DBG & 2) != 0) return;
        if (pkind[k] == 0) {
            const int c = sC + lro + pj[k] * 8;
            const bool w = c >= Wp;
            const int cc = w ? c - Wp : c;
            const bool ok = pch[k] & (lro + pj[k] * 8 < rem) & ((unsigned)(cc - 1) < (unsigned)Wo);
            const int off = pcol[k] + lrA + baseA + pj[k] * 8 * rowA - (w ? 2 * rowA : 0);
            glds16(srcG, ok ? (unsigned)off : OOB, sbase + pdst[k]);
        } else if (pkind[k] == 1) {
            const int c = xC + lro + pj[k] * 8;
            const bool w = c >= Wp;
            const int cc = w ? c - Wp : c;
            const bool ok = pch[k] & (lro + pj[k] * 8 < XROWS) & ((unsigned)(cc - 1) < (unsigned)Wo) & (w ? ok1 : ok0);
            const int off = pcol[k] + lrB + baseB + pj[k] * 8 * pixB - (w ? 2 * pixB : 0);
            glds16(srcX, ok ? (unsigned)off : OOB, sbase + pdst[k]);
        } else {
            glds16(srcG, OOB, dummy);
        }
    };
    auto advance = [&]() {
        sq += PK;
        sC += PK;
        if (sC >= Wp) { sC -= Wp; ++sR; }
        xC += PK;
        if (xC >= Wp) { xC -= Wp; ++xR; xOh = xOh + 1 == Ho ? 0 : xOh + 1; }
    };
    auto issue = [&](int is) {
        prep(is);
#pragma unroll
        for (int k = 0; k < PPW; ++k) piece(k);
        advance();
    };

    // ---- read side: transposed 4 x 16 reads (ds_read_b64_tr_b16) of this wave's gy sub-tile qa and x sub-tile qb
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g4 >> 1;
    const int colw = (16 * (g4 & 1) + 4 * pp) * 2;
    int offA[2], offB[3][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) offA[i] = opaque((8 * hh + q) * 128 + ((i * 64 + colw) ^ (((q >> 1) & 1) << 6)));
#pragma unroll
    for (int kj = 0; kj < 3; ++kj)
#pragma unroll
        for (int j = 0; j < 2; ++j) offB[kj][j] = opaque((8 * hh + q + kj) * 128 + ((j * 64 + colw) ^ ((((q + kj) >> 1) & 1) << 6)));
    const char *ringp = lds_all + grp * NS * SLOT;
    const int rdA = qa * NA * 1024, rdB = A * NA * 1024 + qb * NB * 1024;

#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);
    int rslot = 0, islot = NS - 1;
    for (int n = 0; n < nst; ++n) {
        wait_vmcnt<(NS - 2) * PPW>();
        __builtin_amdgcn_s_barrier();                  // everyone's pieces of stage n have landed; stage n - 1 is consumed
        prep(islot);                                   // stage n + NS - 1 (out of range past the end: zeros), a piece per MFMA group
        const char *pa = ringp + rslot * SLOT + rdA, *pb = ringp + rslot * SLOT + rdB;
#pragma unroll
        for (int ks = 0; ks < PK / 16; ++ks) {
            uint4 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr ((DBG & 8) != 0) { fa[i] = make_uint4(lane, n, ks, i); continue; }
                const char *base = pa + offA[i] + ks * 16 * 128;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * 128));
                uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) {
                uint4 fb[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr ((DBG & 8) != 0) { fb[j] = make_uint4(lane, n, kj, j); continue; }
                    const char *base = pb + offB[kj][j] + ks * 16 * 128;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * 128));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr ((DBG & 1) != 0) acc[kj][i][j][0] += __builtin_bit_cast(float, fa[i].x ^ fb[j].y);
                        else Mma<T>::run(fb[j], fa[i], acc[kj][i][j]);       // D[ci][co]: a lane's 4 consecutive registers = 4 consecutive ci
                    }
                if (ks * 3 + kj < PPW) piece(ks * 3 + kj);
            }
            if (do_sum) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned w[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float lo, hi; unpack2<T>(w[e], lo, hi); fsum[i] += lo + hi; }
                }
            }
        }
        advance();
        rslot = rslot + 1 == NS ? 0 : rslot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }
    wait_vmcnt<0>();                                   // the trailing out-of-range pieces still target this workgroup's LDS
    __syncthreads();                                   // every wave is done with the rings

    // dbeta partial sums: one row per (slab, group); the rows 4 slab + g, g >= G, of the four the ABI promises are zeros
    if (do_sum) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float tot = fsum[i] + __shfl_xor(fsum[i], 32, 64);
            const int co = co0 + qa * 64 + i * 32 + (lane & 31);
            if (lane < 32 && co < a.Cout) {
                a.gsum[(size_t)(slab_id * 4 + grp) * a.Cout + co] = tot;
#pragma unroll
                for (int g = grp + G; g < 4; g += G) a.gsum[(size_t)(slab_id * 4 + g) * a.Cout + co] = 0.f;
            }
        }
    }
    // Cross-group reduction, one tap at a time, in the fixed order group 0 + group 1 + ...; group 0 stores the slab tile.
    float *slab = a.slabs + (size_t)slab_id * a.Cout * 9 * a.Cin;
    if constexpr ((DBG & 4) != 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q2 = 0; q2 < 16; ++q2) t += acc[k][i][j][q2];
        if (t == 1.2345f) slab[0] = 0.f;
        return;
    }
    const int r = lane & 31, h = lane >> 5;
    float4 *red4 = reinterpret_cast<float4 *>(lds_all);
#pragma unroll
    for (int kj = 0; kj < 3; ++kj) {
        if (G > 1) {
            if (kj) __syncthreads();
            if (grp > 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const f32x16 &v = acc[kj][i][j];
                            red4[((((grp - 1) * NQ + wq) * 4 + (i * 2 + j)) * 4 + c4) * 64 + lane] = make_float4(v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]);
                        }
            }
            __syncthreads();
        }
        if (grp == 0) {
            const int tap = ki * 3 + kj;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int co = co0 + qa * 64 + i * 32 + r;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        const f32x16 &v = acc[kj][i][j];
                        float4 sum = make_float4(v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]);
#pragma unroll
                        for (int g2 = 1; g2 < G; ++g2) {
                            const float4 o = red4[((((g2 - 1) * NQ + wq) * 4 + (i * 2 + j)) * 4 + c4) * 64 + lane];
                            sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
                        }
                        const int ci = ci0 + qb * 64 + j * 32 + 8 * c4 + 4 * h;       // registers 4 c4 .. 4 c4 + 3 are columns ci .. ci + 3
                        if (co < a.Cout && ci < a.Cin) *reinterpret_cast<float4 *>(slab + ((size_t)co * 9 + tap) * a.Cin + ci) = sum;
                    }
                }
        }
    }
}

#define DCF_WS_GROUP 32
struct WsGroup {
    WsArgs a[DCF_WS_GROUP];
    int off[DCF_WS_GROUP + 1];
    int n;
};

template <typename T, int A, int BC, int G, int NS>
__global__ void __launch_bounds__(A * BC * G * 64) k_conv_wgrad3s_grp(WsGroup g)
{
    int i = 0;
#pragma unroll
    for (int k = 1; k < DCF_WS_GROUP; ++k) i += (k < g.n && (int)blockIdx.x >= g.off[k]);
    wgs_body<T, A, BC, G, NS>(g.a[i], (int)blockIdx.x - g.off[i]);
}

}  // namespace

int dcf_wgrad_upx(int tiles2, int nsplit);       // conv.hip

// ---- host side (called from conv.hip)
// kind: 0 = not this kernel's; 1 = quadrants 2 x 2 (128 x 128 output tiles: channel counts multiples of 128)
int dcf_wgrad3s_kind(int dtype, int B, int H, int W, int Cin, int Cout)
{
    static DcfOpt e_o("WGRAD3S"); const char *e = e_o.str();
    if (e && atoi(e) == 0) return 0;
    if (dtype == DCF_F32 || W + 2 < 40) return 0;
    if ((int64_t)B * H * W * Cin * 2 >= (1ll << 31) || (int64_t)B * H * W * Cout * 2 >= (1ll << 31)) return 0;
    // (a 192 x 64 tile of 3 x 1 quadrants for the 192-channel layers was built and measured: six waves on four SIMDs, 560-700
    // TFLOP/s in the grouped launch against 680-750 for k_conv_wgrad3g on the same layers -- removed)
    return (Cin % 128 == 0 && Cout % 128 == 0) ? 1 : 0;
}

// pixel ranges (= slabs) of a layer: a workgroup should walk >= ~2048 padded positions (64 stages shared by its two groups), and a
// layer should not need more than ~96 workgroups (the grouped launch overlaps the layers)
int dcf_wgrad3s_splits(int kind, int B, int H, int W, int Cin, int Cout)
{
    static DcfOpt e_o("WGRAD3S_BLOCKS"); const char *e = e_o.str();
    const int want_blocks = e ? atoi(e) : 72;
    const int qco = kind == 1 ? 128 : 192, qci = kind == 1 ? 128 : 64;
    const int tiles = cdiv(Cout, qco) * cdiv(Cin, qci) * 3;
    const int64_t M = (int64_t)B * H * (W + 2);
    int64_t want = std::max<int64_t>(1, want_blocks / tiles);
    const int64_t maxs = std::max<int64_t>(1, M / 2048);
    if (want > maxs) want = maxs;
    return (int)want;
}

struct dcf_wgs_item {
    const void *x, *gy;
    float *slabs, *gsum;
    int B, H, W, Cin, Cout, nsplit;
};

int dcf_wgrad3s_launch(int dtype, int kind, const dcf_wgs_item *items_in, int n, double flops, double bytes, hipStream_t s)
{
    WsGroup g;
    const int qco = kind == 1 ? 128 : 192, qci = kind == 1 ? 128 : 64;
    // longest workgroups first (list schedule over the CUs); equal lengths keep the caller's order
    std::vector<dcf_wgs_item> sorted(items_in, items_in + n);
    std::stable_sort(sorted.begin(), sorted.end(), [](const dcf_wgs_item &p, const dcf_wgs_item &q) {
        return (int64_t)p.B * p.H * (p.W + 2) * q.nsplit > (int64_t)q.B * q.H * (q.W + 2) * p.nsplit;
    });
    const dcf_wgs_item *items = sorted.data();
    for (int i0 = 0; i0 < n; i0 += DCF_WS_GROUP) {
        const int cnt = std::min(DCF_WS_GROUP, n - i0);
        int blocks = 0, rot = 0;
        for (int k = 0; k < cnt; ++k) {
            const dcf_wgs_item &it = items[i0 + k];
            if (((uintptr_t)it.slabs & 15) || (it.Cin & 3)) return DCF_EINVAL;      // 16-byte slab stores
            WsArgs &a = g.a[k];
            a.x = (const char *)it.x; a.gy = (const char *)it.gy; a.slabs = it.slabs; a.gsum = it.gsum;
            a.B = it.B; a.H = it.H; a.W = it.W; a.Cin = it.Cin; a.Cout = it.Cout;
            a.M = it.B * it.H * (it.W + 2);
            a.nsplit = it.nsplit;
            a.per_split = cdiv(cdiv(a.M, it.nsplit), 64) * 64;
            a.co_tiles = cdiv(it.Cout, qco); a.ci_tiles = cdiv(it.Cin, qci);
            a.xbytes = (unsigned)((int64_t)it.B * it.H * it.W * it.Cin * 2);
            a.gbytes = (unsigned)((int64_t)it.B * it.H * it.W * it.Cout * 2);
            g.off[k] = blocks;
            const int units = a.co_tiles * a.ci_tiles * it.nsplit, upx = a.upx = dcf_wgrad_upx(a.co_tiles * a.ci_tiles, it.nsplit);
            blocks += 8 * 3 * upx;
            a.rot = rot;
            rot = (rot + cdiv(units, upx)) & 7;
        }
        static DcfOpt dbg_o("WGRAD3S_DBG"); const char *dbg = dbg_o.str();
        if (dbg && atoi(dbg))
            for (int k = 0; k < cnt; ++k)
                fprintf(stderr, "wgrad3s kind %d: B %d H %d W %d Cin %d Cout %d nsplit %d per_split %d blocks %d\n", kind, g.a[k].B, g.a[k].H,
                        g.a[k].W, g.a[k].Cin, g.a[k].Cout, g.a[k].nsplit, g.a[k].per_split, (k + 1 < cnt ? g.off[k + 1] : blocks) - g.off[k]);
        for (int k = cnt; k <= DCF_WS_GROUP; ++k) g.off[k] = blocks;
        for (int k = cnt; k < DCF_WS_GROUP; ++k) g.a[k] = g.a[0];
        g.n = cnt;
        const double f = flops * cnt / n, by = bytes * cnt / n;       // a launch's share when a bucket spills into several
#define DCF_WGS_GO(A_, B_, G_, NS_)                                                                                                               \
    do {                                                                                                                                          \
        if (dtype == DCF_F16)                                                                                                                     \
            DCF_LAUNCH_WB("conv_wgrad3s_grp_f16<" #A_ "," #B_ "," #G_ "," #NS_ ">", f, by, s,                                                     \
                          hipLaunchKernelGGL((k_conv_wgrad3s_grp<f16_t, A_, B_, G_, NS_>), dim3(blocks), dim3(A_ * B_ * G_ * 64), 0, s, g));       \
        else                                                                                                                                      \
            DCF_LAUNCH_WB("conv_wgrad3s_grp_bf16<" #A_ "," #B_ "," #G_ "," #NS_ ">", f, by, s,                                                    \
                          hipLaunchKernelGGL((k_conv_wgrad3s_grp<bf16_t, A_, B_, G_, NS_>), dim3(blocks), dim3(A_ * B_ * G_ * 64), 0, s, g));      \
    } while (0)
        // (a ring of 4 slots, and one pixel group per workgroup with two workgroups per CU, were measured: no faster)
        DCF_WGS_GO(2, 2, 2, 3);
#undef DCF_WGS_GO
    }
    return DCF_OK;
}
