// conv_wgv.hip -- weight gradient of the 3x3 / stride-1 / pad-1 convolutions with 32 channels on both sides (16-bit types,
// gfx950): LiDAR stage 1 at 704 x 800, the layers whose weight gradient is pure memory traffic (AI 145 flop/B: x + gy = 144 MB
// per cfg2 layer for 41 GFLOP).
//
// k_conv_wgrad3<T, 1, 1, 3> (conv.hip) walks the zero-padded image position by position and stages, for every 32 positions,
// the gy tile and the x tiles of the THREE kernel rows: an x row is fetched three times, one image row (51 KB of x alone) apart
// -- long evicted from every cache in between.  rocprofv3 counted 292 MB fetched per launch for 144 MB of x + gy.
//
// Here a wave walks a 32-column STRIP of the image downwards.  The x tile of input row r + 1 is staged once and stays in the
// wave's LDS ring while it serves kernel row 2 of output row r, kernel row 1 of output row r + 1 and kernel row 0 of output
// row r + 2; the three horizontal taps are the same staged tile read 0 / 1 / 2 pixels further right (34 staged columns), as
// in the row-sharing kernels.  Per output row a wave stages one gy tile (32 px x 64 B) and ONE x tile (34 px x 64 B):
// x is fetched 1.06x (strip halo) + 2 rows per row range, gy once.
//
//   * unit = (frame, strip, range of RR output rows); one wave per unit, four waves per workgroup, one workgroup per CU
//     (LDS: 4 x 26 KB); units are numbered strip-fastest, so the waves of a workgroup share their halo columns.
//   * wave-private rings filled by LDS-DMA (inline asm `buffer_load ... lds`, out-of-range offsets = the zero padding and the
//     image border), three output rows ahead; every step issues the same five pieces (rows past the range as out-of-range
//     pieces), so the loop's only wait is `s_waitcnt vmcnt(15)`; no barrier in the loop.
//   * 9 accumulator tiles (32 x 32, one per tap) per wave; the four waves of a workgroup meet in LDS in fixed order and write
//     ONE fp32 slab (nsplit = workgroups), so dcf_wgrad_finalize and the bitwise reproducibility of the weight gradients
//     are unchanged; d(beta) sums come from the gy fragments the waves hold anyway (4 rows of gsum per slab).
// Algorithmic work: 2*B*H*W*32*32*9 flop; B*H*W*64*2 B read once, nsplit slabs of 36 KB written.
#include <stdlib.h>

#include <algorithm>

#include "dcf_common.h"
#include "conv_common.h"

// timing ablations, compile time only (-DDCF_WGV_DBG_MASK=n): 1 no MFMA, 2 no DMA (out-of-range pieces only), 4 no epilogue
#ifndef DCF_WGV_DBG_MASK
#define DCF_WGV_DBG_MASK 0
#endif

namespace {

struct WvArgs {
    const char *x, *gy;
    float *slabs, *gsum;
    int B, H, W;
    int strips, RR, nrr, units;   // 32-column strips, rows per range, ranges per image, units = B * nrr * strips
    int nwg;                      // workgroups that have work (= slabs)
    unsigned xbytes, gbytes;
};

constexpr int WV_D = 3;                       // output rows staged ahead
constexpr int WV_NX = WV_D + 3, WV_NG = WV_D + 1;
constexpr int WV_XS = 3 * 1024, WV_GS = 2 * 1024;          // slot bytes: 48 (34 used) / 32 rows of 64 B
constexpr int WV_WAVE = WV_NX * WV_XS + WV_NG * WV_GS;     // 26 KB
constexpr int WV_GRP = 5;                     // DMA pieces per step: 2 (gy) + 3 (x)

template <typename T>
__global__ void __launch_bounds__(256) k_conv_wgrad3v(WvArgs a)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    constexpr int C = 32, RB = 64;                // channels, row bytes = LDS pitch
    __shared__ __attribute__((aligned(1024))) char lds_all[4 * WV_WAVE];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ldsX = lds_all + wid * WV_WAVE, *ldsG = ldsX + WV_NX * WV_XS;
    const unsigned ldsX0 = lds_addr(ldsX), ldsG0 = lds_addr(ldsG);

    // XCD-aware order (workgroups are dealt round-robin over the 8 XCDs): XCD x takes the x-th contiguous run of workgroups,
    // so neighbouring strips / row ranges (shared halo columns and rows) meet in one L2
    const int chunk = gridDim.x >> 3;             // the grid is nwg rounded up to a multiple of 8
    const int wg = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (wg >= a.nwg) return;
    const int unit = wg * 4 + wid;
    const bool live = unit < a.units;
    int b = 0, rr = 0, strip = 0;
    if (live) { strip = unit % a.strips; const int t = unit / a.strips; rr = t % a.nrr; b = t / a.nrr; }
    const int c0 = strip * 32;
    const int r0 = rr * a.RR, r1 = live ? min(r0 + a.RR, a.H) : r0;
    const int nsteps = r1 - r0;

    f32x16 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float fsum = 0.f;

    // ---- DMA side: lane = (staged row lr of a 16-row piece, 16-B chunk ch)
    const __amdgpu_buffer_rsrc_t srcG = __builtin_amdgcn_make_buffer_rsrc((void *)a.gy, 0, a.gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int lr = lane >> 2, ch = lane & 3;
    const bool dma_on = !(DCF_WGV_DBG_MASK & 2);
    // per-lane column validity and byte offsets inside an image row (fixed over the walk)
    bool okG[2], okX[3];
    int offG[2], offX[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = c0 + 16 * j + lr;
        okG[j] = dma_on && col < a.W;
        offG[j] = col * RB + ch * 16;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int s = 16 * j + lr, col = c0 - 1 + s;
        okX[j] = dma_on && s < 34 && col >= 0 && col < a.W;
        offX[j] = col * RB + ch * 16;
    }
    const int rowbytes = a.W * RB;
    const int img = b * a.H;
    auto issue_x = [&](int row, int slot) {          // input row `row` of the frame (may be -1 or H: zeros)
        const bool okr = live && (unsigned)row < (unsigned)a.H && row <= r1;
        const int base = __builtin_amdgcn_readfirstlane((img + row) * rowbytes);
        const unsigned dst = __builtin_amdgcn_readfirstlane(ldsX0 + slot * WV_XS);
#pragma unroll
        for (int j = 0; j < 3; ++j) glds16(srcX, (okr && okX[j]) ? (unsigned)(base + offX[j]) : OOB, dst + j * 1024);
    };
    auto issue_g = [&](int row, int slot) {
        const bool okr = row < r1;
        const int base = __builtin_amdgcn_readfirstlane((img + row) * rowbytes);
        const unsigned dst = __builtin_amdgcn_readfirstlane(ldsG0 + slot * WV_GS);
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(srcG, (okr && okG[j]) ? (unsigned)(base + offG[j]) : OOB, dst + j * 1024);
    };

    // ---- read side: lane-constant byte offsets of the transposed fragment reads (64-byte rows, as k_conv_wgrad3g's)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g >> 1;
    const int colw = (16 * (g & 1) + 4 * pp) * 2;
    const int offA = opaque((8 * hh + q) * RB + colw);
    int offB[3];
#pragma unroll
    for (int kj = 0; kj < 3; ++kj) offB[kj] = opaque((8 * hh + q + kj) * RB + colw);

    // x rows are numbered j = 0 .. nsteps + 1 (input row r0 - 1 + j, ring slot j % NX); step n (output row r0 + n) reads
    // j = n, n + 1, n + 2 and gy row n (slot n % NG).  Issue order: X(0) X(1) | G(0) X(2) | G(1) X(3) | ...
    issue_x(r0 - 1, 0);
    issue_x(r0, 1);
#pragma unroll
    for (int s = 0; s < WV_D; ++s) { issue_g(r0 + s, s); issue_x(r0 + s + 1, s + 2); }
    int xs0 = 0;                      // slot of x row j = n
    int gs = 0;                       // slot of gy row n
    int xi = (WV_D + 2) % WV_NX;      // slot the next x row goes to
    int gi = WV_D % WV_NG;
    for (int n = 0; n < nsteps; ++n) {
        issue_g(r0 + n + WV_D, gi);
        issue_x(r0 + n + WV_D + 1, xi);
        wait_vmcnt<WV_D * WV_GRP>();
        const char *pg = ldsG + gs * WV_GS;
        const char *px[3];
        {
            int xs = xs0;
#pragma unroll
            for (int ki = 0; ki < 3; ++ki) { px[ki] = ldsX + xs * WV_XS; xs = xs + 1 == WV_NX ? 0 : xs + 1; }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fa;
            {
                const char *base = pg + offA + ks * 16 * RB;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * RB));
                uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fa = make_uint4(l2.x, l2.y, h2.x, h2.y);
            }
            if (a.gsum != nullptr) {
                const unsigned w[4] = {fa.x, fa.y, fa.z, fa.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { float lo, hi; unpack2<T>(w[e], lo, hi); fsum += lo + hi; }
            }
#pragma unroll
            for (int ki = 0; ki < 3; ++ki)
#pragma unroll
                for (int kj = 0; kj < 3; ++kj) {
                    const char *base = px[ki] + offB[kj] + ks * 16 * RB;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(base + 4 * RB));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    const uint4 fb = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    if (!(DCF_WGV_DBG_MASK & 1)) Mma<T>::run(fa, fb, acc[ki][kj]);
                }
        }
        // the reads above must have returned before the next step's DMA may overwrite their slots (the MFMAs that consume
        // them are not ordered against an asm statement by data flow)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        xs0 = xs0 + 1 == WV_NX ? 0 : xs0 + 1;
        gs = gs + 1 == WV_NG ? 0 : gs + 1;
        xi = xi + 1 == WV_NX ? 0 : xi + 1;
        gi = gi + 1 == WV_NG ? 0 : gi + 1;
    }
    wait_vmcnt<0>();                                   // out-of-range tail pieces still write (zeros) into the rings
    __syncthreads();                                   // every wave is done with its ring
    const int slab_id = wg;
    if (a.gsum != nullptr) {
        const float tot = fsum + __shfl_xor(fsum, 32, 64);
        if (lane < 32) a.gsum[(size_t)(slab_id * 4 + wid) * C + lane] = tot;
    }
    if (DCF_WGV_DBG_MASK & 4) return;
    // Cross-wave reduction, three taps (one kernel row) at a time: every wave parks its tiles in LDS, then wave w sums float4
    // group w of the four copies in the fixed order w0 + w1 + w2 + w3 and stores it.
    float *slab = a.slabs + (size_t)slab_id * C * 9 * C;
    const int r = lane & 31, h = lane >> 5;
    float4 *red4 = reinterpret_cast<float4 *>(lds_all);       // [kj 3][wave 4][group 4][lane 64] float4 = 48 KB
#pragma unroll
    for (int ki = 0; ki < 3; ++ki) {
        if (ki) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#pragma unroll
        for (int kj = 0; kj < 3; ++kj)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x16 &v = acc[ki][kj];
                red4[((kj * 4 + wid) * 4 + g4) * 64 + lane] = make_float4(v[4 * g4], v[4 * g4 + 1], v[4 * g4 + 2], v[4 * g4 + 3]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int kj = 0; kj < 3; ++kj) {
            const int g4 = wid;
            float4 sum = red4[((kj * 4 + 0) * 4 + g4) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 v = red4[((kj * 4 + w) * 4 + g4) * 64 + lane];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const int tap = ki * 3 + kj;
            const int ci = r;
            const int co = 8 * g4 + 4 * h;                     // registers 4*g4 .. 4*g4+3 are rows co .. co+3
            float *dst = slab + ((size_t)co * 9 + tap) * C + ci;
            const size_t rs = (size_t)9 * C;
            dst[0] = sum.x; dst[rs] = sum.y; dst[2 * rs] = sum.z; dst[3 * rs] = sum.w;
        }
    }
}

void wgv_plan(int B, int H, int W, WvArgs &a)
{
    a.strips = cdiv(W, 32);
    static DcfOpt u_o("WGRAD3V_UNITS");
    const int want = u_o.str() ? std::max(atoi(u_o.str()), 1) : 1024;       // one wave per unit, 4 waves per CU
    int nrr = std::max(1, (want + B * a.strips / 2) / (B * a.strips));
    nrr = std::min(nrr, std::max(1, H / 8));                             // >= 8 rows per range (2 halo rows each)
    a.RR = cdiv(H, nrr);
    a.nrr = cdiv(H, a.RR);
    a.units = B * a.nrr * a.strips;
}

}  // namespace

// 0 = not this kernel's layer, else the number of slabs (= workgroups) it writes for the shape
int dcf_wgrad3v_splits(int B, int H, int W, int Cin, int Cout)
{
    static DcfOpt off_o("WGRAD3V");
    if (off_o.str() && atoi(off_o.str()) == 0) return 0;
    if (Cin != 32 || Cout != 32 || H < 8 || W < 8) return 0;
    if ((int64_t)B * H * W * 64 >= (1ll << 31)) return 0;
    WvArgs a;
    wgv_plan(B, H, W, a);
    return cdiv(a.units, 4);
}

int dcf_wgrad3v_launch(int dtype, const void *x, const void *gy, float *slabs, float *gsum, int nsplit, int B, int H, int W, double flops,
                       hipStream_t s)
{
    WvArgs a;
    wgv_plan(B, H, W, a);
    DCF_REQUIRE(nsplit == cdiv(a.units, 4), "dcf_conv2d_wgrad: nsplit %d does not match dcf_conv2d_wgrad_splits (%d) for this layer", nsplit, cdiv(a.units, 4));
    a.x = (const char *)x; a.gy = (const char *)gy; a.slabs = slabs; a.gsum = gsum;
    a.B = B; a.H = H; a.W = W; a.nwg = nsplit;
    const int grid = cdiv(nsplit, 8) * 8;
    a.xbytes = a.gbytes = (unsigned)((int64_t)B * H * W * 64);
    const double bytes = 2.0 * a.xbytes + (double)nsplit * 32 * 9 * 32 * 4.0;
    if (dtype == DCF_F16) DCF_LAUNCH_WB("conv_wgrad3v_f16", flops, bytes, s, hipLaunchKernelGGL(k_conv_wgrad3v<f16_t>, dim3(grid), dim3(256), 0, s, a));
    else DCF_LAUNCH_WB("conv_wgrad3v_bf16", flops, bytes, s, hipLaunchKernelGGL(k_conv_wgrad3v<bf16_t>, dim3(grid), dim3(256), 0, s, a));
    return DCF_OK;
}
