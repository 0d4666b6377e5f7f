// conv_wg1.hip -- weight gradient with SHARED staging for the convolutions the row-sharing kernels do not take: 1x1 at any
// stride (shortcuts, FPN laterals, the fusion MLPs, ResNet-50 bottlenecks) and 3x3 / stride 2 / pad 1 (the first convolution of
// every down-sampling stage: /root/reference/model.py:15-28 with stride 2, :30 the 1x1 shortcut).  16-bit types, gfx950.
//
// Until round 5 these layers ran on the generic kernel of conv.hip (k_conv_wgrad): every wave stages its OWN 32-pixel tiles
// through registers for eight MFMAs -- 32 flop per loaded byte, 230-270 TFLOP/s in the grouped launch (0.10 of the MFMA peak;
// 0.19 ms of a cfg2 step, 0.97 ms of a cfg4 step, where it was the top kernel).  Halving that launch's HBM traffic (XCD-aware
// mapping) moved it 3 %: what it waits for is its staging.  Here, as in conv_wgs.hip:
//   * the eight waves of a workgroup are two groups of four 64 x 64 QUADRANTS of one 128 (Cout) x 128 (Cin) tile of ONE tap;
//     a group's waves share every staged 32-pixel stage: 8 gy pieces + 8 x pieces of 1 KiB for 32 MFMAs (64 flop per staged
//     byte), LDS-DMA ring of NS stages per group, one barrier per stage, counted vmcnt;
//   * wave w of a group issues piece w of each of the stage's four sub-tiles -- rows 8 w .. 8 w + 7 -- so a lane follows ONE
//     output pixel (frame, row, column advanced by 32 per stage: no division in the loop) and derives from it the gy row and,
//     for tap (ki, kj), the input pixel (oh s + ki - pad, ow s + kj - pad): any stride, image borders = out-of-range DMA = zeros;
//   * the two groups take the two halves of the workgroup's pixel range and are summed through LDS in the fixed order
//     group 0 + group 1 -> one fp32 slab tile per workgroup, weight gradients bitwise reproducible;
//   * grid = (pixel range, channel-tile pair) units x taps, XCD-aware: the taps of a unit run on one XCD (they share gy and
//     overlap in x), an XCD owns a contiguous run of units.
// With one tap per stage the loop is bound by what a CU takes in through LDS-DMA (32 KiB per pair of group stages at ~23 B/clk
// against 512 MFMA cycles per SIMD): ~0.35 of the MFMA peak at best.
// Algorithmic work per layer: 2*B*Ho*Wo*Cout*Cin*kh*kw flop; x and gy read once, nsplit fp32 slab sets written.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "dcf_common.h"
#include "conv_common.h"

namespace {

struct W1Args {
    const char *x;    // [B][H][W][Cin]
    const char *gy;   // [B][Ho][Wo][Cout]
    float *slabs;     // [nsplit][Cout][taps][Cin]
    float *gsum;      // [4*nsplit][Cout] or null
    int B, H, W, Cin, Ho, Wo, Cout;
    int kh, kw, stride, pad;
    int M;            // output pixels B*Ho*Wo
    int nsplit, per_split;   // pixels per workgroup (multiple of 64)
    int co_tiles, ci_tiles;
    unsigned xbytes, gbytes;
    int rot;          // XCD of the layer's first run of units
    int upx;          // units per XCD
};

// HA / HB: the workgroup's tile has a second 64-channel sub-tile on the Cout / Cin side.  A sub-tile the layer does not have (64
// channels on a side, or the last tile of 192) is neither staged nor multiplied: 12 or 8 KiB per stage instead of 16, the waves of
// the missing quadrants only issue their share of the pieces and keep the barriers.
// LD (round 5, last change): SIXTEEN waves -- waves 0-7 are the consumers (fragment reads, MFMAs, reduction, stores; no DMA piece),
// waves 8-15 the loaders (wave 8 + w issues exactly the pieces wave w issued and nothing else).  As in conv_rs_kernel.h: a wave
// stalled in the vector-memory queue issues no MFMAs, and this kernel's waves spent half their cycles parked at the fill.
template <typename T, int NS, bool HA, bool HB, bool LD>
__device__ __forceinline__ void wg1_tile(const W1Args &a, const int unit, const int tap, char *lds_all)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    constexpr int G = 2, NQ = 4, PK = 32;
    constexpr int NP = 16;                              // 1-KiB pieces (8 rows x 128 B) per stage and group: 2 gy + 2 x sub-tiles x 4
    constexpr int PPW = 2 + (HA ? 1 : 0) + (HB ? 1 : 0);   // pieces per wave and stage: piece `wq` of every sub-tile the tile has
    constexpr int SLOT = NP * 1024;
    constexpr int RING = G * NS * SLOT;
    constexpr int RED = (G - 1) * NQ * 16 * 1024;       // the 64 x 64 tile of every wave of group 1
    static_assert(RING >= RED && RING <= 160 * 1024, "LDS (wg1_body declares the ring; the reduction re-uses it)");
    static_assert((NS - 1) * PPW < 60, "vmcnt range");
    const int lane = threadIdx.x & 63;
    const int wid16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = LD && wid16 >= G * NQ;          // (wave-uniform)
    const int wid = loader ? wid16 - G * NQ : wid16;
    const int grp = wid / NQ, wq = wid - grp * NQ;
    const int qa = wq >> 1, qb = wq & 1;                // this wave's quadrant: co sub-tile qa, ci sub-tile qb
    const unsigned lds0 = lds_addr(lds_all);
    const unsigned ring0 = lds0 + grp * NS * SLOT;

    const int taps = a.kh * a.kw;
    const int tiles2 = a.co_tiles * a.ci_tiles;
    const int ki = tap / a.kw, kj = tap - ki * a.kw;
    const int slab_id = unit / tiles2;
    const int t2 = unit - slab_id * tiles2;
    const int cit = t2 % a.ci_tiles, cot = t2 / a.ci_tiles;
    const int co0 = cot * 128, ci0 = cit * 128;
    const int gspan = a.per_split / G;
    const int q_begin = slab_id * a.per_split + grp * gspan;
    const int q_end = min(min(q_begin + gspan, slab_id * a.per_split + a.per_split), a.M);
    const int nst = gspan / PK;                          // the same for both groups: the barriers line up

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const bool active = !loader && (HA || qa == 0) && (HB || qb == 0);     // this wave computes a quadrant that exists
    const bool issues = LD ? loader : true;                                 // this wave issues DMA pieces
    const bool do_sum = active && (a.gsum != nullptr) && (tap == 0) && (cit == 0) && (qb == 0);
    float fsum[2] = {0.f, 0.f};

    // ---- DMA side.  Lane = (row lr of the 8-row piece, 16-B chunk); 128-B rows, 64-B halves swapped on odd row pairs (the four
    // rows of a transposed read then sit on distinct banks); the swizzle goes on the SOURCE chunk (conv_wgs.hip).
    const int rowA = a.Cout * 2, pixB = a.Cin * 2;
    const int lr = lane >> 3;
    const int ch = (lane & 7) ^ (((lr >> 1) & 1) << 2);
    const __amdgpu_buffer_rsrc_t srcG = __builtin_amdgcn_make_buffer_rsrc((void *)a.gy, 0, a.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    // which 64-channel sub-tiles exist / which 16-B chunks of them are inside the tensors
    bool chA[2], chB[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        chA[s] = ch * 16 < min(64, a.Cout - (co0 + s * 64)) * 2;
        chB[s] = ch * 16 < min(64, a.Cin - (ci0 + s * 64)) * 2;
    }
    const int colA = co0 * 2 + ch * 16, colB = ci0 * 2 + ch * 16;
    // this lane's output pixel: row 8 wq + lr of the stage
    int p = q_begin + 8 * wq + lr;
    int pb, poh, pow_;
    {
        const int hw = a.Ho * a.Wo;
        pb = p / hw;
        const int r = p - pb * hw;
        poh = r / a.Wo; pow_ = r - poh * a.Wo;
    }
    const int dh = ki - a.pad, dw = kj - a.pad;
    auto issue = [&](int is) {
        if (!issues) return;
        const unsigned sbase = __builtin_amdgcn_readfirstlane(ring0 + is * SLOT) + (unsigned)wq * 1024u;
        const bool live = p < q_end;
        const int offA = p * rowA + colA;
        const int ih = poh * a.stride + dh, iw = pow_ * a.stride + dw;
        const bool inimg = live & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        const int offB = ((pb * a.H + ih) * a.W + iw) * pixB + colB;
        glds16(srcG, (live & chA[0]) ? (unsigned)offA : OOB, sbase);
        if constexpr (HA) glds16(srcG, (live & chA[1]) ? (unsigned)(offA + 128) : OOB, sbase + 4 * 1024);
        glds16(srcX, (inimg & chB[0]) ? (unsigned)offB : OOB, sbase + 8 * 1024);
        if constexpr (HB) glds16(srcX, (inimg & chB[1]) ? (unsigned)(offB + 128) : OOB, sbase + 12 * 1024);
        p += PK;
        pow_ += PK;
        while (pow_ >= a.Wo) { pow_ -= a.Wo; ++poh; }
        while (poh >= a.Ho) { poh -= a.Ho; ++pb; }
    };

    // ---- read side: transposed 4 x 16 reads (ds_read_b64_tr_b16) of this wave's gy sub-tile qa and x sub-tile qb
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g4 >> 1;
    const int colw = (16 * (g4 & 1) + 4 * pp) * 2;
    int offR[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) offR[i] = opaque((8 * hh + q) * 128 + ((i * 64 + colw) ^ (((q >> 1) & 1) << 6)));
    const char *ringp = lds_all + grp * NS * SLOT;
    const int rdA = qa * 4 * 1024, rdB = 8 * 1024 + qb * 4 * 1024;

#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);
    int rslot = 0, islot = NS - 1;
    for (int n = 0; n < nst; ++n) {
        wait_vmcnt<(NS - 2) * PPW>();
        __builtin_amdgcn_s_barrier();                  // everyone's pieces of stage n have landed; stage n - 1 is consumed
        issue(islot);                                  // stage n + NS - 1 (past the end of the range: zeros)
        const char *pa = ringp + rslot * SLOT + rdA, *pbk = ringp + rslot * SLOT + rdB;
        if (active)
#pragma unroll
        for (int ks = 0; ks < PK / 16; ++ks) {
            uint4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char *base = pa + offR[i] + ks * 16 * 128;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * 128));
                uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char *base = pbk + offR[j] + ks * 16 * 128;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * 128));
                uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<T>::run(fb[j], fa[i], acc[i][j]);       // D[ci][co]: a lane's 4 consecutive registers = 4 consecutive ci
            if (do_sum) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned w[4] = {fa[i].x, fa[i].y, fa[i].z, fa[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float lo, hi; unpack2<T>(w[e], lo, hi); fsum[i] += lo + hi; }
                }
            }
        }
        rslot = rslot + 1 == NS ? 0 : rslot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }
    wait_vmcnt<0>();                                   // the trailing out-of-range pieces still target this workgroup's LDS
    __syncthreads();                                   // every wave is done with the rings

    // dbeta partial sums: one row per (slab, group); rows 4 slab + 2, + 3 of the four the ABI promises are zeros
    if (do_sum) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float tot = fsum[i] + __shfl_xor(fsum[i], 32, 64);
            const int co = co0 + qa * 64 + i * 32 + (lane & 31);
            if (lane < 32 && co < a.Cout) {
                a.gsum[(size_t)(slab_id * 4 + grp) * a.Cout + co] = tot;
                a.gsum[(size_t)(slab_id * 4 + grp + G) * a.Cout + co] = 0.f;
            }
        }
    }
    // cross-group reduction in the fixed order group 0 + group 1; group 0 stores the slab tile
    float *slab = a.slabs + (size_t)slab_id * a.Cout * taps * a.Cin;
    const int r = lane & 31, h = lane >> 5;
    float4 *red4 = reinterpret_cast<float4 *>(lds_all);
    if (grp > 0 && active) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const f32x16 &v = acc[i][j];
                    red4[((wq * 4 + (i * 2 + j)) * 4 + c4) * 64 + lane] = make_float4(v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]);
                }
    }
    __syncthreads();
    if (grp == 0 && active) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = co0 + qa * 64 + i * 32 + r;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const f32x16 &v = acc[i][j];
                    const float4 o = red4[((wq * 4 + (i * 2 + j)) * 4 + c4) * 64 + lane];
                    const float4 sum = make_float4(v[4 * c4] + o.x, v[4 * c4 + 1] + o.y, v[4 * c4 + 2] + o.z, v[4 * c4 + 3] + o.w);
                    const int ci = ci0 + qb * 64 + j * 32 + 8 * c4 + 4 * h;       // registers 4 c4 .. 4 c4 + 3 are rows ci .. ci + 3
                    if (co < a.Cout && ci < a.Cin) *reinterpret_cast<float4 *>(slab + ((size_t)co * taps + tap) * a.Cin + ci) = sum;
                }
            }
    }
}

template <typename T, int NS, bool LD>
__device__ __forceinline__ void wg1_body(const W1Args &a, const int bid)
{
    // XCD-aware unit order: workgroup bid runs on XCD bid & 7; an XCD owns a contiguous run of (range, co tile, ci tile) units x taps
    const int taps = a.kh * a.kw;
    const int tiles2 = a.co_tiles * a.ci_tiles;
    const int units = tiles2 * a.nsplit, upx = a.upx;       // whole ranges per XCD (dcf_wgrad_upx, conv.hip)
    const int slot_id = bid >> 3;
    const int unit = (((bid & 7) - a.rot) & 7) * upx + slot_id / taps;
    if (unit >= units) return;
    __shared__ __attribute__((aligned(1024))) char lds_all[2 * NS * 16 * 1024];      // two groups' rings of NS 16-KiB stages
    const int tap = slot_id % taps;
    const int t2 = unit % tiles2;
    const bool ha = a.Cout - (t2 / a.ci_tiles) * 128 > 64, hb = a.Cin - (t2 % a.ci_tiles) * 128 > 64;      // workgroup-uniform
    if (ha) {
        if (hb) wg1_tile<T, NS, true, true, LD>(a, unit, tap, lds_all); else wg1_tile<T, NS, true, false, LD>(a, unit, tap, lds_all);
    } else {
        if (hb) wg1_tile<T, NS, false, true, LD>(a, unit, tap, lds_all); else wg1_tile<T, NS, false, false, LD>(a, unit, tap, lds_all);
    }
}

#define DCF_W1_GROUP 32
struct W1Group {
    W1Args a[DCF_W1_GROUP];
    int off[DCF_W1_GROUP + 1];
    int n;
};

template <typename T, int NS, bool LD = false>
__global__ void __launch_bounds__(LD ? 1024 : 512) k_conv_wgrad1s_grp(W1Group g)
{
    int i = 0;
#pragma unroll
    for (int k = 1; k < DCF_W1_GROUP; ++k) i += (k < g.n && (int)blockIdx.x >= g.off[k]);
    wg1_body<T, NS, LD>(g.a[i], (int)blockIdx.x - g.off[i]);
}

}  // namespace

int dcf_wgrad_upx(int tiles2, int nsplit);       // conv.hip

// ---- host side (called from conv.hip)
// 1 = this kernel's layer: 16-bit, 64-channel granules on both sides (a stage row of a sub-tile is 128 B), 1x1 / pad 0 at any stride
// or 3x3 / stride 2 / pad 1, enough output pixels for a pipeline (smaller layers stay on the generic kernel).  Option WGRAD1S=0: never.
int dcf_wgrad1s_kind(int dtype, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad)
{
    static DcfOpt e_o("WGRAD1S"); const char *e = e_o.str();
    if (e && atoi(e) == 0) return 0;
    static DcfOpt g_o("WGRAD1S_GRANULE"); const char *gr = g_o.str();
    const int gran = gr ? atoi(gr) : 64;                 // 128: only layers that fill the 128 x 128 tile
    if (dtype == DCF_F32 || Cin % gran || Cout % gran || Cin % 64 || Cout % 64) return 0;
    const bool k1 = kh == 1 && kw == 1 && pad == 0 && stride >= 1;
    const bool k3 = kh == 3 && kw == 3 && stride == 2 && pad == 1;
    if (!k1 && !k3) return 0;
    if ((int64_t)B * H * W * Cin * 2 >= (1ll << 31) || (int64_t)B * Ho * Wo * Cout * 2 >= (1ll << 31)) return 0;
    static DcfOpt c_o("WGRAD1S_MIN_CH"); const char *mc = c_o.str();
    if (std::min(Cin, Cout) < (mc ? atoi(mc) : 64)) return 0;
    static DcfOpt m_o("WGRAD1S_MIN_PIXELS"); const char *m = m_o.str();
    if ((int64_t)B * Ho * Wo < (m ? atoi(m) : 2048)) return 0;
    return 1;
}

// pixel ranges (= slabs) of a layer: ~WGRAD1S_BLOCKS workgroups per layer (the grouped launch overlaps the layers), at least 1024
// pixels (16 stages per group) per workgroup, at most 16 MB of slabs
int dcf_wgrad1s_splits(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw)
{
    static DcfOpt e_o("WGRAD1S_BLOCKS"); const char *e = e_o.str();
    const int want_blocks = e ? atoi(e) : 144;
    const int tiles = cdiv(Cout, 128) * cdiv(Cin, 128) * kh * kw;
    const int64_t M = (int64_t)B * Ho * Wo;
    int64_t want = std::max<int64_t>(1, want_blocks / tiles);
    const int64_t maxs = std::max<int64_t>(1, M / 1024);
    if (want > maxs) want = maxs;
    const int64_t slab_bytes = (int64_t)Cout * kh * kw * Cin * 4;
    static DcfOpt cap_env_o("SLAB_CAP_MB"); const char *cap_env = cap_env_o.str();
    const int64_t cap = std::max<int64_t>(1, ((int64_t)(cap_env ? atoi(cap_env) : 16) << 20) / slab_bytes);
    if (want > cap) want = cap;
    return (int)want;
}

struct dcf_wg1_item {                 // (mirrors the declaration in conv.hip)
    const void *x, *gy;
    float *slabs, *gsum;
    int B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, nsplit;
};

int dcf_wgrad1s_launch(int dtype, const dcf_wg1_item *items_in, int n, double flops, double bytes, hipStream_t s)
{
    W1Group g;
    // longest workgroups first (list schedule over the CUs); equal lengths keep the caller's order
    std::vector<dcf_wg1_item> sorted(items_in, items_in + n);
    std::stable_sort(sorted.begin(), sorted.end(), [](const dcf_wg1_item &p, const dcf_wg1_item &q) {
        return (int64_t)p.B * p.Ho * p.Wo * q.nsplit > (int64_t)q.B * q.Ho * q.Wo * p.nsplit;
    });
    const dcf_wg1_item *items = sorted.data();
    const int ns = 4;        // (a ring of 3 stages measured the same on cfg4: 0.711 against 0.722 ms; one instantiation is kept)
    for (int i0 = 0; i0 < n; i0 += DCF_W1_GROUP) {
        const int cnt = std::min(DCF_W1_GROUP, n - i0);
        int blocks = 0, rot = 0;
        for (int k = 0; k < cnt; ++k) {
            const dcf_wg1_item &it = items[i0 + k];
            if (((uintptr_t)it.slabs & 15) || (it.Cin & 3) || it.nsplit < 1) return DCF_EINVAL;      // 16-byte slab stores
            W1Args &a = g.a[k];
            a.x = (const char *)it.x; a.gy = (const char *)it.gy; a.slabs = it.slabs; a.gsum = it.gsum;
            a.B = it.B; a.H = it.H; a.W = it.W; a.Cin = it.Cin; a.Ho = it.Ho; a.Wo = it.Wo; a.Cout = it.Cout;
            a.kh = it.kh; a.kw = it.kw; a.stride = it.stride; a.pad = it.pad;
            a.M = it.B * it.Ho * it.Wo;
            a.nsplit = it.nsplit;
            a.per_split = cdiv(cdiv(a.M, it.nsplit), 64) * 64;
            a.co_tiles = cdiv(it.Cout, 128); a.ci_tiles = cdiv(it.Cin, 128);
            a.xbytes = (unsigned)((int64_t)it.B * it.H * it.W * it.Cin * 2);
            a.gbytes = (unsigned)((int64_t)it.B * it.Ho * it.Wo * it.Cout * 2);
            g.off[k] = blocks;
            const int units = a.co_tiles * a.ci_tiles * it.nsplit, upx = a.upx = dcf_wgrad_upx(a.co_tiles * a.ci_tiles, it.nsplit);
            blocks += 8 * it.kh * it.kw * upx;
            a.rot = rot;
            rot = (rot + cdiv(units, upx)) & 7;
        }
        for (int k = cnt; k <= DCF_W1_GROUP; ++k) g.off[k] = blocks;
        for (int k = cnt; k < DCF_W1_GROUP; ++k) g.a[k] = g.a[0];
        g.n = cnt;
        const double f = flops * cnt / n, by = bytes * cnt / n;       // a launch's share when a bucket spills into several
#define DCF_WG1_GO(NS_, LD_, TH_)                                                                                                                 \
    do {                                                                                                                                          \
        if (dtype == DCF_F16)                                                                                                                     \
            DCF_LAUNCH_WB("conv_wgrad1s_grp_f16<" #NS_ ">", f, by, s, hipLaunchKernelGGL((k_conv_wgrad1s_grp<f16_t, NS_, LD_>), dim3(blocks), dim3(TH_), 0, s, g)); \
        else                                                                                                                                      \
            DCF_LAUNCH_WB("conv_wgrad1s_grp_bf16<" #NS_ ">", f, by, s, hipLaunchKernelGGL((k_conv_wgrad1s_grp<bf16_t, NS_, LD_>), dim3(blocks), dim3(TH_), 0, s, g)); \
    } while (0)
        (void)ns;
        static DcfOpt ld_o("WGRAD1S_L16"); const char *lde = ld_o.str();
        // eight consumer + eight loader waves (default) or the eight-wave form (WGRAD1S_L16=0)
        if (lde && atoi(lde) == 0) DCF_WG1_GO(4, false, 512); else DCF_WG1_GO(4, true, 1024);
#undef DCF_WG1_GO
    }
    return DCF_OK;
}
