// conv_sp.hip -- spatial-tile streaming kernel for the HBM-bound 3x3 / stride-1 / pad-1 convolutions with 32 or 64 channels on
// both sides (forward and input gradient), 16-bit element types, gfx950.
//
// These are the LiDAR stream's first two stages (/root/reference/model.py:15-28 ResidualBlock bodies of layer1 / layer2:
// 32 channels at 704x800, 64 at 352x400): 72-144 MB in and out per launch against 10-20 GFLOP -- arithmetic intensity 145-290
// flop/B, below the chip's ridge, so what bounds them is how few bytes move and how well the moves overlap.  The implicit-GEMM
// kernels walk K tap by tap and stage a pixel once per tap (conv.hip) or once per kernel row (conv_rs.hip): 9x / 3x the
// tensor through the L2 -> LDS path, and each workgroup's fetch latency, nine tap steps and store burst serialise (measured
// 2.5-2.8 TB/s, EXPERIMENTS.md, round 3).  Here:
//
//   * ONE staging per pixel.  A workgroup owns a TH x 32 output tile of one frame and DMAs the (TH + 2) x 34 input pixels
//     (`buffer_load ... lds`, image borders and tile tails as out-of-range offsets = zeros) into LDS once; all nine taps read
//     that tile at pixel offsets (ki, kj).  Staged bytes per output byte: (TH + 2) * 34 / (TH * 32) = 1.33 at TH = 8; the halo
//     rows mostly hit the XCD's L2 (neighbouring tiles run on the same XCD at about the same time).
//   * WEIGHTS IN REGISTERS.  9 * C / 16 MFMA A-fragments per 32 output channels (18 for C = 32, 36 for C = 64) are loaded once
//     per (persistent) workgroup and stay in VGPRs: the main loop's only LDS traffic is one ds_read_b128 per MFMA, its only
//     global traffic the tile DMA, the residual / mask vectors and the stores.
//   * PERSISTENT workgroups, two per CU, each a ring of NBUF = 3 staged tiles: the DMA of tile t + 2 goes out BEFORE the MFMAs
//     of tile t, so a tile's fetch latency (~2.4 us per iteration even with nothing to compute, tools/sp_ablate.py -- several
//     times its MFMA time) is spread over two iterations and 88-104 KB per CU are in flight (a one-tile-ahead version moved
//     52-66 KB per 2.4 us per CU = the 2.4 TB/s it measured on the 64-channel layers).  Every wave issues the same number of
//     DMA pieces per tile (surplus ones out of range into a dump slot) and the same number of stores (unconditional buffer
//     stores, tile tails = out-of-range offsets), so the wait at the top of the loop is a counted one that leaves the later
//     tiles' pieces and the previous tiles' stores in flight (they retire in issue order, MI355X_MICROARCH.md; as in
//     conv_rs.hip).  One barrier per tile.
//   * dgrad is the same kernel on the [Cin][tap][Cout] weight image with the taps mirrored (as conv_rs.hip).
//
// Algorithmic work per launch: 2*B*H*W*C*C*9 flop; bytes B*H*W*C*2 * (2 + residual + mask) + weights.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "dcf_common.h"
#include "conv_common.h"

namespace {

struct SpArgs {
    const char *x;        // [B][H][W][C]
    const char *w;        // [C][9][C]
    const float *shift;   // [C] or null
    const char *res;      // [B*H*W][C] or null
    const char *mask;     // [B*H*W][C] or null: output *= (mask > 0)
    char *y;              // [B*H*W][C]
    int B, H, W;
    int relu, flip;       // flip = 1: input gradient (taps mirrored)
    int txn, tyn;         // tiles per row / per frame column
    int ntiles;
    unsigned xbytes;
    int dbg;              // timing ablations, -DDCF_ABLATE builds only (option SP_DBG): 1 no LDS reads / MFMAs, 2 DMA reads nothing, 4 no stores
};

typedef unsigned sp_u32x4 __attribute__((ext_vector_type(4)));

// C channels (= Cin = Cout), TH output rows per tile; 4 waves: WN = C / 32 channel tiles x WM = 4 / WN row groups of TM = TH / WM rows.
//
// Instruction budget.  tools/sp_ablate.py on a first version: with the MFMAs, the fetched data and the stores all switched off the
// launch still took 17-19 of its 33 us -- the per-tile INSTRUCTION stream (a dozen VALU operations per DMA piece for its pixel
// coordinates, three per LDS read for its swizzled address), not latency.  So: (1) the staged row pitch is a multiple of 256
// bytes (34 pixels of 128 B, 36 of 64 B) and the swizzle key is a function of the staged COLUMN only, which makes a lane's LDS
// read address base(row group) + table(kj, ks) + an immediate (ki): one v_add per read, the 3 x KS table entries live in
// registers; (2) a lane's source offset of piece j relative to the tile's origin is tile-invariant: kept in a register, an
// interior tile's DMA address is one v_add (border tiles recompute coordinates and range checks).
template <typename T, int C, int TH, int NBUF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NBUF == 2 ? 3 : 2, NBUF == 2 ? 3 : 2))) k_conv3x3_sp(SpArgs a)
{
    static_assert(DT<T>::size == 2, "16-bit element types only");
    static_assert(C == 32 || C == 64, "32 or 64 channels");
    constexpr int TW = 32;                              // tile width
    constexpr int PW = C == 64 ? 34 : 36;               // staged width: pitch PW * RB is a multiple of 256 B
    constexpr int RB = C * 2;                           // bytes per pixel
    constexpr int NPX = (TH + 2) * PW;                  // staged pixels
    constexpr int PPP = 1024 / RB;                      // pixels per DMA piece (64 lanes x 16 B)
    constexpr int NPIECE = (NPX + PPP - 1) / PPP;
    constexpr int PXW = (NPIECE + 3) / 4;               // pieces a wave issues per tile
    constexpr int WN = C / 32, WM = 4 / WN, TM = TH / WM;
    constexpr int KS = C / 16;                          // MFMA k-steps per tap
    constexpr int NST = TM * 2;                         // store instructions per wave and tile
    static_assert(TH % WM == 0 && (PW * RB) % 256 == 0, "rows must split evenly over the waves; row pitch");
    constexpr int DUMP = NBUF * NPIECE;                 // LDS slot of the surplus (out-of-range) pieces
    constexpr int AHEAD = (NBUF - 2) * (PXW + NST) + NST;   // operations a wave issues after the pieces of the tile it is about to read
    static_assert(NBUF >= 2 && AHEAD < 64, "vmcnt range");
    static_assert((NBUF * NPIECE + 1) * 1024 * (NBUF == 2 ? 3 : 2) <= 160 * 1024, "LDS: two workgroups per CU (three with a two-slot ring)");
    __shared__ __attribute__((aligned(1024))) char lds[(NBUF * NPIECE + 1) * 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid % WN, wm = wid / WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware persistent tile walk: XCD x = blockIdx & 7 owns the x-th contiguous chunk of the tile list (tiles of a row
    // adjacent, rows of a frame consecutive), its workgroups take the chunk's tiles round-robin.
    const int chunk = (a.ntiles + 7) >> 3;
    const int wpx = gridDim.x >> 3;
    const int chunk_lo = (blockIdx.x & 7) * chunk, chunk_hi = min(chunk_lo + chunk, a.ntiles);
    int t = chunk_lo + (blockIdx.x >> 3);
    if (t >= chunk_hi) return;

    const __amdgpu_buffer_rsrc_t srcX = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dstY = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.xbytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const unsigned lds0 = lds_addr(lds);

    // ---- weights -> registers: fragment (tap, ks) of this wave's 32 output channels.  Lane (r, h) holds w[n0 + r][tap][16 ks + 8 h .. + 8).
    uint4 wreg[9 * KS];
    {
        const char *wp = a.w + (size_t)(wn * 32 + r) * (9 * RB) + h * 16;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int ts = a.flip ? 8 - tp : tp;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[tp * KS + ks] = *reinterpret_cast<const uint4 *>(wp + ts * RB + ks * 32);
        }
    }

    // ---- DMA side.  Piece p covers staged pixels PPP p .. PPP p + PPP - 1 (pixel i = staged row i / PW, column i % PW); lane l ->
    // pixel PPP p + l / CPP, LDS chunk position l % CPP, which holds source chunk position ^ key(column): key = (cc >> 1) & 7
    // for 128-byte pixels, (cc >> 2) & 3 for 64-byte ones -- the 16 consecutive pixels of a ds_read_b128 lane group then sit on
    // distinct banks at every tap offset (the row pitch being a multiple of 256 B, a row's bank phase is that of row 0).
    constexpr int CPP = RB / 16;                                   // 16-byte chunks per pixel
    auto colkey = [](int cc) { return C == 64 ? (cc >> 1) & 7 : (cc >> 2) & 3; };
    int relo[PXW];              // byte offset of this lane's source chunk of piece j from the tile's first staged pixel; < 0: none
#pragma unroll
    for (int j = 0; j < PXW; ++j) {
        const int p = wid + 4 * j;
        const int i = p * PPP + lane / CPP;
        const int rr_ = i / PW, cc_ = i - rr_ * PW;
        relo[j] = (p < NPIECE && i < NPX) ? (rr_ * a.W + cc_) * RB + (((lane % CPP) ^ colkey(cc_)) << 4) : -1;
    }
    auto issue_tile = [&](int tile, int buf) __attribute__((always_inline)) {
        const int tx = tile % a.txn, tq = tile / a.txn;
        const int ty = tq % a.tyn, b = tq / a.tyn;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        const bool inner = tile < chunk_hi && y0 >= 0 && y0 + TH + 2 <= a.H && x0 >= 0 && x0 + PW <= a.W;      // wave-uniform
        const int org = ((b * a.H + y0) * a.W + x0) * RB;
        if (inner) {
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                const int p = wid + 4 * j;                          // (p >= NPIECE: a surplus piece, so that every wave issues PXW per tile)
                glds16(srcX, (relo[j] >= 0 && !(DCF_DBG(a) & 2)) ? (unsigned)(org + relo[j]) : OOB,
                       __builtin_amdgcn_readfirstlane(lds0 + (p < NPIECE ? buf * NPIECE + p : DUMP) * 1024));
            }
        } else {
            const int ln = opaque(lane);                            // (recomputed, not kept: border tiles are the few)
#pragma unroll
            for (int j = 0; j < PXW; ++j) {
                const int p = wid + 4 * j;
                const int i = p * PPP + ln / CPP;
                const int rr_ = i / PW, cc_ = i - rr_ * PW;
                const int yy = y0 + rr_, xx = x0 + cc_;
                const bool live = tile < chunk_hi && relo[j] >= 0 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                glds16(srcX, (live && !(DCF_DBG(a) & 2)) ? (unsigned)(org + relo[j]) : OOB,
                       __builtin_amdgcn_readfirstlane(lds0 + (p < NPIECE ? buf * NPIECE + p : DUMP) * 1024));
            }
        }
    };

    // ---- read side.  Staged pixel of (tile row R, tap ki kj) for this lane = (R + ki) PW + r + kj; its k-step ks is the 16 bytes at
    // chunk position (2 ks + h) ^ key(r + kj): byte address = [R PW RB + r RB] + [kj RB + (((2 ks + h) ^ key(r + kj)) << 4)] + ki PW RB
    //                                                        = rowbase(R)      + xoff[kj][ks]                              + immediate
    int xoff[3][KS];
#pragma unroll
    for (int kj = 0; kj < 3; ++kj)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xoff[kj][ks] = (r + kj) * RB + (((2 * ks + h) ^ colkey(r + kj)) << 4);

    const T *res = reinterpret_cast<const T *>(a.res);
    const T *mask = reinterpret_cast<const T *>(a.mask);
    const int c0 = wn * 32 + 8 * h;

    // prologue: tiles t .. t + (NBUF - 2) wpx, each followed by stand-ins for a tile's stores: from here on the wave's stream of
    // operations is [PXW pieces][NST stores] repeated, whatever the tile (past the chunk's end: out-of-range pieces)
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d) {
        issue_tile(t + d * wpx, d);
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const sp_u32x4 z = {0u, 0u, 0u, 0u};
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(z), "v"(OOB), "s"(dstY) : "memory");
        }
    }
    int buf = 0;
    for (;;) {
        const int tx = t % a.txn, tq = t / a.txn;
        const int ty = tq % a.tyn, b = tq / a.tyn;
        const int tn = t + wpx;
        wait_vmcnt<AHEAD>();                                        // this tile's pieces have landed; later tiles' pieces and earlier stores may still fly
        __builtin_amdgcn_s_barrier();                               // ... everyone's; and everyone is done with the buffer of the previous tile
        issue_tile(t + (NBUF - 1) * wpx, buf == 0 ? NBUF - 1 : buf - 1);      // NBUF - 1 tiles ahead, into the buffer just read
        const char *tile = lds + buf * (NPIECE * 1024) + wm * TM * PW * RB;
        f32x16 acc[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
        if (!(DCF_DBG(a) & 1))
#pragma unroll
        for (int ki = 0; ki < 3; ++ki)
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) {
                asm volatile("" ::: "memory");       // keep a tap's LDS reads behind the previous tap's: hoisting all nine costs ~100 VGPRs
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    uint4 fb[TM];
#pragma unroll
                    for (int j = 0; j < TM; ++j) fb[j] = *reinterpret_cast<const uint4 *>(tile + xoff[kj][ks] + (j + ki) * PW * RB);
#pragma unroll
                    for (int j = 0; j < TM; ++j) Mma<T>::run(wreg[(ki * 3 + kj) * KS + ks], fb[j], acc[j]);
                }
            }
        // ---- epilogue: v = acc + shift + res ; relu ; v *= (mask > 0), 8 consecutive channels per access
        int m[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int yy = ty * TH + wm * TM + j, xx = tx * TW + r;
            m[j] = (yy < a.H && xx < a.W) ? (b * a.H + yy) * a.W + xx : -1;
            acc_rows8(acc[j]);
        }
        if (res) {
            uint4 rr[TM][2];
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    rr[j][p] = m[j] >= 0 ? *reinterpret_cast<const uint4 *>(res + (size_t)m[j] * C + c0 + 16 * p) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const unsigned rw[4] = {rr[j][p].x, rr[j][p].y, rr[j][p].z, rr[j][p].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        unpack2<T>(rw[e], lo, hi);
                        acc[j][8 * p + 2 * e] += lo; acc[j][8 * p + 2 * e + 1] += hi;
                    }
                }
        }
        if (a.shift) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float sh[8];
                ld8(a.shift + c0 + 16 * p, sh);
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[j][8 * p + k] += sh[k];
            }
        }
        if (a.relu) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[j][k] = fmaxf(acc[j][k], 0.f);
        }
        if (mask) {
            uint4 mm[TM][2];
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    mm[j][p] = m[j] >= 0 ? *reinterpret_cast<const uint4 *>(mask + (size_t)m[j] * C + c0 + 16 * p) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const unsigned mw[4] = {mm[j][p].x, mm[j][p].y, mm[j][p].z, mm[j][p].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        unpack2<T>(mw[e], lo, hi);
                        if (!(lo > 0.f)) acc[j][8 * p + 2 * e] = 0.f;
                        if (!(hi > 0.f)) acc[j][8 * p + 2 * e + 1] = 0.f;
                    }
                }
        }
        // exactly NST store instructions per wave, whatever the tile: lanes without an output pixel store out of range (= nothing)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = acc[j][8 * p + k];
                uint4 pk;
                if constexpr (std::is_same<T, bf16_t>::value) {
                    pk = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
                } else {
                    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
                    h16x8 hv;
#pragma unroll
                    for (int k = 0; k < 8; ++k) hv[k] = (_Float16)v[k];
                    pk = __builtin_bit_cast(uint4, hv);
                }
                const sp_u32x4 d = {pk.x, pk.y, pk.z, pk.w};
                __builtin_amdgcn_raw_buffer_store_b128(d, dstY, (m[j] >= 0 && !(DCF_DBG(a) & 4)) ? (int)((unsigned)(m[j] * C + c0 + 16 * p) * 2u) : (int)OOB, 0, 0);
            }
        if (tn >= chunk_hi) break;
        t = tn;
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
}

}  // namespace

// Called by dcf_conv2d_fwd / dcf_conv2d_dgrad (conv.hip) before the row-sharing kernel.  Returns DCF_EUNSUPPORTED when the
// shape is not this kernel's: Cin == Cout in {32, 64}, a 16-bit type, and enough pixels that the launch is HBM-bound (option
// CONV_SP_MIN_PIX, default 200 000 = LiDAR stages 1-2 at every batch size; below that the launch is latency-bound and the
// row-sharing kernel's bigger tiles win).
int dcf_conv3x3_sp_launch(int dtype, const void *x, const void *w, const float *shift, const void *res, const void *mask, void *y,
                          int B, int H, int W, int Ck, int Cn, int relu, int flip, const char *name_base, double flops, hipStream_t s)
{
    static DcfOpt on_o("CONV_SP"), minpix_o("CONV_SP_MIN_PIX"), wgpc_o("CONV_SP_WGPC");
    const char *on = on_o.str(), *mp = minpix_o.str(), *wg = wgpc_o.str();
    if (on && atoi(on) == 0) return DCF_EUNSUPPORTED;
    if (dtype == DCF_F32 || Ck != Cn || (Ck != 32 && Ck != 64)) return DCF_EUNSUPPORTED;
    const int64_t npix = (int64_t)B * H * W;
    if (npix < (mp ? atoll(mp) : 200000ll) || npix * Ck * 2 >= (1ll << 31)) return DCF_EUNSUPPORTED;
    const int TH = Ck == 32 ? 8 : 4;                               // rows per tile (both: two rows per wave; 22 / 26 KB per staged tile)
    SpArgs a;
    a.x = (const char *)x; a.w = (const char *)w; a.shift = shift; a.res = (const char *)res; a.mask = (const char *)mask; a.y = (char *)y;
    a.B = B; a.H = H; a.W = W; a.relu = relu; a.flip = flip;
    a.txn = (W + 31) / 32; a.tyn = (H + TH - 1) / TH;
    a.ntiles = B * a.txn * a.tyn;
    a.xbytes = (unsigned)(npix * Ck * 2);
    a.dbg = dcf_ablate_opt("SP_DBG");
    static DcfOpt nb_o("CONV_SP_NBUF32");
    const char *nb = nb_o.str();
    const bool ring2 = Ck == 32 && !(nb && atoi(nb) == 3);          // 32 channels: two-slot rings, three workgroups per CU
    const int wgpc = wg ? atoi(wg) : (ring2 ? 3 : 2);              // workgroups per CU the LDS rings allow
    const int64_t nwg = std::min<int64_t>(((int64_t)a.ntiles + 7) / 8 * 8, 256 * wgpc);
    const dim3 grid((unsigned)nwg);
    char name[96];
    snprintf(name, sizeof(name), "%s<sp%d>", name_base, Ck);
    const double bytes = (double)npix * Ck * 2.0 * (2 + (res ? 1 : 0) + (mask ? 1 : 0)) + 9.0 * Ck * Cn * 2.0;
#define DCF_SP(T_)                                                                                                                \
    do {                                                                                                                          \
        if (Ck == 32 && ring2) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_sp<T_, 32, 8, 2>), grid, dim3(256), 0, s, a)); \
        else if (Ck == 32) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_sp<T_, 32, 8, 3>), grid, dim3(256), 0, s, a)); \
        else DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_sp<T_, 64, 4, 3>), grid, dim3(256), 0, s, a));       \
    } while (0)
    if (dtype == DCF_F16) DCF_SP(f16_t); else DCF_SP(bf16_t);
#undef DCF_SP
    return DCF_OK;
}
