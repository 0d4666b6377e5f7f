// conv_rs_kernel.h -- kernel template shared by conv_rs.hip (the single-layer launches) and conv_chain.hip (the chain
// launches): two translation units, so that the two sets of instantiations compile side by side.
//
// Row-sharing implicit GEMM -- row-sharing implicit GEMM for the 3x3 / stride-1 / pad-1 convolutions (forward and input gradient),
// 16-bit element types, gfx950.
//
// These are 47 of the network's convolutions (/root/reference/model.py:15-28 ResidualBlock bodies, :153 conv3; the
// camera trunk's BasicBlocks): ~1.0 TFLOP of the ~1.2 TFLOP a cfg2 forward + dgrad needs.  The generic implicit GEMM
// (conv.hip) stages, per tap and 64-channel chunk, one weight tile and one pixel tile: 32-64 flop per staged byte, and
// what a CU can pull out of its XCD's L2 into LDS (~70 GB/s, MI355X_MICROARCH.md "Indexed rows: gather into LDS") caps
// such a kernel at 25-45 % of the MFMA rate.  Here the staged bytes per flop drop 1.5-2.4x:
//
//   * ROW SHARING.  Output pixels are enumerated over the zero-PADDED image (rows of W + 2 positions; positions 0 and
//     W + 1 of every row are padding: their outputs are dropped, their inputs read as zero).  In that flattened space the
//     input of tap (ki, kj) for position p is position p + (ki - 1)(W + 2) + (kj - 1), so the three horizontal taps of a
//     kernel row read ONE staged pixel tile of BM + 2 rows at row offsets 0, 1, 2 -- an x tile is staged once per
//     (kernel row, channel chunk) instead of once per tap.
//   * BIG, SHAPED TILES.  One 512-thread workgroup per CU; a tile is BN = 128 or 64 channels x BM = 32 * npt positions
//     with npt chosen per layer on the host so that the tile count fills the 256 CUs in whole rounds (the generic
//     kernel's 550 tiles of 128x128 on the 128-channel stage leave a third round 15 % full).
//   * LDS-DMA rings, one barrier per tap.  `buffer_load ... lds` (inline asm, counted vmcnt) fills a 3-deep weight ring
//     (one slot per tap) two taps ahead and a 2-deep pixel ring (one slot per kernel row x chunk) one stage ahead; padding,
//     tile tails and image borders are out-of-range buffer offsets for which the DMA writes zeros.  Per tap:
//     counted wait -> s_barrier -> issue -> MFMAs.
//   * dgrad is the same kernel on the [Cin][tap][Cout] weight image with the taps mirrored.
//
// Algorithmic work per launch: 2*B*H*W*Cout*Cin*9 flop; bytes B*H*W*(Cin + Cout)*2 (+ residual / mask reads).
#pragma once
#include <stddef.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "dcf_common.h"
#include "conv_common.h"

#ifndef DCF_RS_SPREAD
#define DCF_RS_SPREAD 1
#endif

namespace {

struct RsArgs {
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
    int dbg;              // experiments (DCF_RS_DBG, tools/rs_ablate.py): 1 no MFMAs, 2 pixel DMA reads nothing, 4 no epilogue, 16 weight DMA reads nothing
};

// ---- CHAIN mode (round 5): several same-shaped 3x3 / stride-1 layers (Ck == Cn), each reading the one before it, in ONE
// persistent launch -- the bodies of a residual stage (/root/reference/model.py:32-41: conv1 -> bn1 -> relu -> conv2 -> bn2 ->
// += shortcut -> relu, block after block) forward, and the same list backwards for the input gradients.  At batch 2 every one
// of these layers is a single round of workgroups of 15-30 us of which ~10 are launch, first cold fill and store drain
// (DESIGN.md "What was not reached"); inside one launch a workgroup goes from its tile of layer l to its tile of layer l + 1
// as soon as the tiles THAT tile reads are complete -- no kernel boundary, no grid-wide wait:
//   * work item (l, t): layer-major; a workgroup owns the same tiles t in every layer (the XCD-aware order of the plain
//     kernel), so it never waits inside a layer and a launch whose workgroups are all resident cannot deadlock (grid <= 256,
//     one workgroup per CU: conv_rs_chain_launch);
//   * hand-off (cdna_hip_programming.md Guideline 16, counter form): output stores are write-through (sc1); every wave drains
//     (vmcnt 0), barrier, ONE lane adds 1 to cnt[l][position tile] (agent scope); the consumer's wave 0 polls the counters of
//     the position tiles its halo touches (relaxed sc1 loads, one per lane, s_sleep between polls) until each holds the number
//     of channel tiles, then a barrier releases the other waves; EVERY load of handed-off bytes is an sc1 load (the pixel
//     LDS-DMA pieces and the residual vectors) -- or, with RsChainArgs::acquire set, one agent-scope acquire follows the poll;
//   * spins are bounded (spin_limit polls): a workgroup that gives up records (layer, tile) in ws[1], stops waiting and the
//     host reports DCF_ELAUNCH-class failure through dcf_conv3x3_chain_status -- wrong results, never a hung GPU;
//   * the counters are zero at every launch: the host zeroes the workspace once when it is created, the workgroup that
//     finishes last (ws[0]) zeroes them again.
// -DRS_STAMP (variant builds only: KFILE=conv_chain bash tools/rw_variants.sh stamp="-DRS_STAMP"; tools/chain_stamps.py): wave 0 of
// every workgroup keeps s_memtime stamps of a work item's phases in scalar registers and writes them out when the item is done:
//   0 item start (before the wait for the previous layer's tiles)   1 wait satisfied   2 first DMA groups issued
//   3 first stage's data landed (first MFMAs)   4 last MFMA issued   5 DMA tail drained   6 epilogue stores issued
//   7 stores drained   8 arrival published
#ifdef RS_STAMP
#define RS_STAMP_WGS 256
#define RS_STAMP_ITEMS 32
__device__ long long g_rs_stamps[RS_STAMP_WGS][RS_STAMP_ITEMS][10];
#define RS_T(k) do { if constexpr (CHAIN) { if (wid == 0) stamp[k] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define RS_T(k) do { } while (0)
#endif
// -DRS_WSTAMP (variant builds only: KFILE=conv_rs bash tools/rw_variants.sh wstamp="-DRS_WSTAMP"; tools/rs_wstamps.py): EVERY wave of a
// single-layer launch keeps the low 32 bits of s_memtime for its arrival at and its release from every tap barrier of its first
// tile, plus six phase marks, in three VGPRs (one lane per stamp: v_writelane -- no memory instruction, so the counted vmcnt waits
// are untouched) and writes them out when the workgroup is done:
//   misc 0 kernel entry   1 first DMA groups + stand-in stores issued   2 last MFMA issued   3 DMA tail drained   4 epilogue stores
//   issued   5 stores drained
#ifdef RS_WSTAMP
#define RS_WS_WGS 256
__device__ unsigned g_rs_wstamps[RS_WS_WGS][8][3][64];
#define RS_WS(reg, idx) do { if constexpr (!CHAIN) { const unsigned t__ = (unsigned)__builtin_amdgcn_s_memtime(); const int i__ = __builtin_amdgcn_readfirstlane(idx); \
    unsigned k__; asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(reg), "=&s"(k__) : "s"(t__), "s"(i__)); } } while (0)
#else
#define RS_WS(reg, idx) do { } while (0)
#endif
#define RS_CHAIN_MAX 24
struct RsChainLayer {
    const char *x, *w;
    const float *shift;
    const char *res, *mask;
    char *y;
    int relu, pad_;
};
struct RsChainArgs {
    RsArgs a;             // geometry of every layer (x / w / shift / res / mask / y / relu unused: per layer below)
    int nlayers;
    int spin_limit;       // polls before a waiting workgroup gives up
    int acquire;          // 1: buffer_inv sc1 after every successful poll (on top of the sc1 loads)
    int pad_;
    int *ws;              // [0] finished workgroups, [1] give-up record, [2..3] unused, [4 + l * mtiles + pt] arrivals
    RsChainLayer L[RS_CHAIN_MAX];
};
// a pointer the compiler must keep in scalar registers (buffer descriptors are "s" operands of the inline-asm DMA): the layer
// table is indexed by a loop-carried variable, whose uniformity the compiler does not always prove
template <typename P> __device__ __forceinline__ P rs_uniform(P p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (P)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ const RsArgs &rs_common(const RsArgs &a) { return a; }
__device__ __forceinline__ const RsArgs &rs_common(const RsChainArgs &c) { return c.a; }

// One output store: 64 lanes x 16 B through a buffer descriptor; a lane whose offset is out of range stores nothing.
// UNCONDITIONAL (padding lanes get an out-of-range offset instead of a branch): the number of store instructions a wave
// issues per tile is then a compile-time constant, which the counted vmcnt waits of the next tile's first steps need -- stores
// count in vmcnt, in issue order, with the DMA pieces (MI355X_MICROARCH.md), and they are left in flight under those steps.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int AUX = 0>          // AUX 16 = sc1: write-through, what a hand-off to another workgroup inside the launch needs (chain mode)
__device__ __forceinline__ void gst16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, uint4 data)
{
    // the compiler's own buffer-store intrinsic, not inline asm: its hazard recogniser then keeps later writes of the data
    // registers away from the store (an asm store followed by `s_nop` was not enough once ten of them went out back to back)
    const u32x4 d = {data.x, data.y, data.z, data.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)voff, 0, AUX);
}
// the LDS-DMA piece of conv_common.h with the sc1 bit: the load bypasses this CU's L1 (served by L2 / the fabric), so bytes
// another workgroup published earlier in this launch are read fresh without an acquire
__device__ __forceinline__ void glds16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}
// A stand-in for a store where a tile has none to issue (the first tile of a workgroup, debug runs): out of range, so nothing is
// written, but it counts in vmcnt like a real one.  Inline asm: ten identical intrinsic stores would be merged into one.
__device__ __forceinline__ void gst16_dummy(__amdgpu_buffer_rsrc_t rsrc)
{
    const u32x4 d = {0u, 0u, 0u, 0u};
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(d), "v"(0xFFFFFF00u), "s"(rsrc) : "memory");
}
template <typename T> __device__ __forceinline__ uint4 pack8(const float (&v)[8]);
template <> __device__ __forceinline__ uint4 pack8<bf16_t>(const float (&v)[8])
{
    return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
template <> __device__ __forceinline__ uint4 pack8<f16_t>(const float (&v)[8])
{
    typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
    h16x8 h;
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = (_Float16)v[k];
    return __builtin_bit_cast(uint4, h);
}

__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// s_waitcnt vmcnt(n) for a wave-uniform runtime n (the immediate must be a constant: one scalar branch)
__device__ __forceinline__ void wait_vmcnt_dyn(int n)
{
    switch (n) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<1>(); break;
    case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;
    case 4: wait_vmcnt<4>(); break;
    case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;
    case 7: wait_vmcnt<7>(); break;
    case 8: wait_vmcnt<8>(); break;
    case 9: wait_vmcnt<9>(); break;
    case 10: wait_vmcnt<10>(); break;
    case 11: wait_vmcnt<11>(); break;
    default: wait_vmcnt<12>(); break;       // n >= 12: waiting for more than necessary is always safe
    }
}

// Instructions a wave has issued AFTER the group that tap kj of a stage depends on (= the N of its s_waitcnt vmcnt(N)):
// group t holds pww weight pieces plus pxa (tap 0) / pxb (tap 1) pixel pieces; a tap depends on the group dw steps back
// (its weights) and tap 0 also on the pixel tile issued with groups 3 dx and 3 dx - 1 steps back.
constexpr int rs_allowed(int kj, int dw, int dx, int pww, int pxa, int pxb)
{
    int back = dw;
    if (kj == 0 && 3 * dx - 1 < back) back = 3 * dx - 1;
    int n = 0;
    for (int d = 1; d < back; ++d) {
        const int k = ((kj - d) % 3 + 3) % 3;
        n += pww + (k == 0 ? pxa : (k == 1 ? pxb : 0));
    }
    return n;
}

// Block = WN x WM waves.  Wave (wn, wm) owns channel tiles wn*TN .. +TN-1 (32 channels each) and its even share of the
// workgroup's npt position tiles (at most TMMAX).  K order: kernel row ki, 64-channel chunk cc, then the three taps.
// DW = taps the weight DMA runs ahead (ring of DW + 1 slots), DX = stages the pixel DMA runs ahead (DX + 1 slots).
// S3: one wait + barrier per STAGE (three taps) instead of per tap -- the small-M layers' tap steps are four MFMAs per wave,
// shorter than the wait / barrier / issue sequence around them.  The weights then run DW = 3 DX taps ahead in a ring of
// DW + 3 tap slots (a stage's three slots are refilled together), the pixel tile DX stages ahead as before.
// CHAIN: the launch walks RsChainArgs::L layer by layer (see RsChainArgs); false = one layer, as before.
// L16 (round 5, option RS_L16, the small-M kind only): SIXTEEN waves -- waves 0 .. NW-1 are CONSUMERS (the tiles, fragment reads,
// MFMAs and the epilogue of the 8-wave form, but no DMA piece), waves NW .. 2 NW - 1 are LOADERS (wave NW + w issues exactly the
// pieces wave w issues in the 8-wave form, right behind each barrier, and nothing else: they are "waves without tiles", the C == 0
// path below).  tools/probe/fill_paths.hip: a CU takes 59-63 B/clk into LDS when eight waves do nothing but issue pieces, the
// 8-wave form reaches 20-23 because a wave stalled in the vector-memory queue issues no MFMAs either.
// NL = number of loader waves (0: the 8-wave form; 8: sixteen waves; 4: twelve waves -- a loader then issues the pieces of TWO of
// the 8-wave form's waves, i.e. the pieces are dealt over NL waves instead of over NW; 170 registers per wave instead of 128, for
// the kinds whose consumers hold 48-80 accumulator registers).
// PF (round 6): the tap loop ROTATED and the fragment reads double-buffered.  The plain loop reads a k-step's fragments, waits for
// them (lgkmcnt 0) and only then issues its MFMAs, and starts every tap behind a barrier with nothing in flight: with two waves per
// SIMD the matrix pipe sat idle for an LDS latency per k-step and for barrier + latency per tap (~50 % busy inside the loop, SQ
// counters of rounds 4-5; the disassembly showed `ds_read ; s_waitcnt lgkmcnt(0) ; v_mfma` pairs).  Here (a) k-step ks + 1's fragments
// are requested BEFORE the MFMAs of k-step ks are issued (two register sets, order pinned with sched_barrier), and (b) the wait +
// barrier of tap t + 1 sits in front of the LAST k-step of tap t, whose MFMAs then cover the first fragment reads of tap t + 1; group
// t + 1's DMA pieces follow that barrier (one per k-step as before).  Every LDS read a wave issued from tap t's slots has returned
// when it arrives there (explicit lgkmcnt(0): the last k-step's fragments are needed at once anyway), so the slot hand-over to the
// DMA is a dependency, not a race the L2 latency wins.  Same groups, same order, same counted vmcnt immediates; bit-identical sums.
template <typename T, int TN, int TMMAX, int WN, int WM, int DW, int DX, bool S3 = false, bool CHAIN = false, int NL = 0, bool PF = false>
__global__ void __launch_bounds__((WN * WM + NL) * 64) k_conv3x3_rs(typename std::conditional<CHAIN, RsChainArgs, RsArgs>::type arg)
{
    constexpr bool L16 = NL > 0;
    static_assert(!PF || !CHAIN, "the rotated loop: single-layer launches");
    static_assert(!(PF && S3) || NL > 0, "the rotated loop with stage-granular synchronisation: the consumer + loader form only");
    static_assert(!(L16 && CHAIN) && (NL == 0 || NL == 4 || NL == 8), "loader waves: none, four or eight; no chain mode with them");
    const RsArgs &a = rs_common(arg);
    static_assert(DT<T>::size == 2, "16-bit element types only");
    constexpr int NW = WN * WM;
    constexpr int BN = WN * TN * 32;
    constexpr int BMMAX = WM * TMMAX * 32;
    constexpr int NWD = NL ? NL : NW;                         // waves the DMA pieces are dealt over
    constexpr int PWW = BN / 8 / NWD;                         // weight pieces (8 rows x 128 B) per issuing wave and tap
    static_assert(PWW >= 1 && PWW * 8 * NWD == BN, "weight tile must split evenly over the issuing waves");
    constexpr int PXW = (BMMAX + 2 + 8 * NWD - 1) / (8 * NWD);  // most pixel pieces an issuing wave issues per stage
    constexpr int XROWS = NWD * PXW * 8;
    constexpr int WSLOT = BN * 128, XSLOT = XROWS * 128;
    constexpr int NSW = S3 ? DW + 3 : DW + 1, NSX = DX + 1;
    static_assert(DW >= 1 && DX >= 1 && (DW - 1) * PWW + 2 * PXW <= 48, "vmcnt range");
    static_assert(!S3 || DW == 3 * DX, "stage-granular sync: weights and pixels run the same number of stages ahead");
    static_assert(NSW * WSLOT + NSX * XSLOT <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char lds[NSW * WSLOT + NSX * XSLOT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid16 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = L16 && wid16 >= WN * WM;           // (wave-uniform)
    const int wid = L16 ? (wid16 >= WN * WM ? wid16 - WN * WM : wid16) : wid16;       // consumer: its tile role; loader: its DMA share
    // Waves w and w + 4 share a SIMD (and its matrix pipe): the second half of the workgroup takes the position shares in
    // reverse order, so that a wave with one tile more is paired with a wave with one tile less.
    const int wn = wid / WM;
    const int wm = (wid & 4) ? WM - 1 - wid % WM : wid % WM;
    const int r = lane & 31, h = lane >> 5;

    // PERSISTENT workgroups, XCD-aware tile order (speed only): XCD x = blockIdx & 7 owns the x-th contiguous chunk of the
    // (position tile, channel tile) list, channel tiles fastest; its workgroups take the chunk's tiles round-robin.  A
    // workgroup issues the first DMA groups of its NEXT tile before it stores the current one, so the L2 latency of a
    // tile's first taps (2-3 us, a third of a 64-channel tile's whole K loop) hides under the previous tile's epilogue.
    const int nt = a.Cn / BN;
    const int nblk = a.mtiles * nt;
    const int chunk = (nblk + 7) >> 3;
    const int wpx = gridDim.x >> 3;                   // workgroups per XCD
    const int chunk_lo = (blockIdx.x & 7) * chunk, chunk_hi = min(chunk_lo + chunk, nblk);
    const int gidx_first = chunk_lo + (blockIdx.x >> 3);
    int gidx = gidx_first;
    // chain mode: the workgroup that finishes last (tile-less ones count too) zeroes the arrival counters for the next launch
    auto chain_finish = [&]() __attribute__((always_inline)) {
        if constexpr (CHAIN) {
            __builtin_amdgcn_s_barrier();
            int *flag = reinterpret_cast<int *>(lds);              // (the rings are idle by now; no second __shared__ object)
            if (tid == 0) *flag = __hip_atomic_fetch_add(arg.ws, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
            __syncthreads();
            if (*flag) {
                for (int i = tid; i < arg.nlayers * a.mtiles; i += NW * 64) arg.ws[4 + i] = 0;
                if (tid == 0) arg.ws[0] = 0;
            }
        }
    };
    if (gidx >= chunk_hi) { chain_finish(); return; }
    int n0 = 0, q0 = 0;
    const int BM = a.npt * 32;
    const int Wp = a.W + 2, BH = a.B * a.H;
    const int rowbytes = a.Ck * 2;
    const int cchunks = rowbytes / 128;
    const int nstage = 3 * cchunks, nsteps = 3 * nstage;

    // this wave's share of the position tiles
    const int base = a.npt / WM, rem = a.npt - base * WM;
    const int cnt = loader ? 0 : base + (wm < rem ? 1 : 0);          // (a loader wave of the 16-wave form has no tiles)
    const int pt0 = wm * base + min(wm, rem);

    // operands of the layer being computed (chain mode: re-bound at every layer switch)
    __amdgpu_buffer_rsrc_t srcX, srcW, dstY, srcR;
    const float *lshift;
    const T *res, *mask;
    int lrelu, layer = 0;
    const char *px, *pw, *pr;
    char *py;
    // (chain mode re-builds the descriptors at the top of every tile from the layer's pointers passed through readfirstlane:
    // the compiler does not prove a loop-carried descriptor uniform, and the inline-asm DMA takes it as an "s" operand)
    auto make_descs = [&]() __attribute__((always_inline)) {
        const char *ux = px, *uw = pw, *ur = pr;
        char *uy = py;
        if constexpr (CHAIN) { ux = rs_uniform(px); uw = rs_uniform(pw); ur = rs_uniform(pr); uy = rs_uniform(py); }
        srcX = __builtin_amdgcn_make_buffer_rsrc((void *)ux, 0, a.xbytes, 0x00020000);
        srcW = __builtin_amdgcn_make_buffer_rsrc((void *)uw, 0, a.wbytes, 0x00020000);
        dstY = __builtin_amdgcn_make_buffer_rsrc((void *)uy, 0, a.ybytes, 0x00020000);
        srcR = __builtin_amdgcn_make_buffer_rsrc((void *)(ur ? ur : uy), 0, a.ybytes, 0x00020000);
    };
    auto bind_layer = [&](int l) __attribute__((always_inline)) {
        const char *pm;
        if constexpr (CHAIN) {
            // the layer table is read straight out of the kernel-argument segment (scalar loads at a uniform offset): indexing
            // the by-value `arg.L[l]` with a run-time l would make the compiler copy the whole 1.6 KB argument to scratch memory
            typedef const __attribute__((address_space(4))) char *kconst_t;
            const kconst_t kp = (kconst_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RsChainArgs, L) +
                                (size_t)__builtin_amdgcn_readfirstlane(l) * sizeof(RsChainLayer);
            const __attribute__((address_space(4))) unsigned long long *q = (const __attribute__((address_space(4))) unsigned long long *)kp;
            static_assert(sizeof(RsChainLayer) == 56 && offsetof(RsChainLayer, y) == 40 && offsetof(RsChainLayer, relu) == 48, "layout read word by word");
            px = (const char *)q[0]; pw = (const char *)q[1]; lshift = (const float *)q[2]; pr = (const char *)q[3]; pm = (const char *)q[4];
            py = (char *)q[5]; lrelu = (int)(unsigned)q[6];
        } else {
            px = a.x; pw = a.w; pr = a.res; pm = a.mask; py = a.y; lshift = a.shift; lrelu = a.relu;
        }
        make_descs();
        res = reinterpret_cast<const T *>(pr);
        mask = reinterpret_cast<const T *>(pm);
    };
    auto bind_weights = [&](int l) __attribute__((always_inline)) {          // chain mode: the weights of layer l only (prefetch)
        if constexpr (CHAIN) {
            typedef const __attribute__((address_space(4))) char *kconst_t;
            const kconst_t kp = (kconst_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RsChainArgs, L) +
                                (size_t)__builtin_amdgcn_readfirstlane(l) * sizeof(RsChainLayer);
            pw = (const char *)((const __attribute__((address_space(4))) unsigned long long *)kp)[1];
            srcW = __builtin_amdgcn_make_buffer_rsrc((void *)rs_uniform(pw), 0, a.wbytes, 0x00020000);
        }
    };
    bind_layer(0);
    constexpr unsigned OOB = 0xFFFFFF00u;
    const unsigned ldsW0 = lds_addr(lds), ldsX0 = ldsW0 + NSW * WSLOT;

    // ---- DMA side.  Lane = (row l8 of the 8-row piece, 16-B chunk position lc); LDS position lc of row R holds source
    // chunk lc ^ ((R >> 1) & 7): the 16 rows of a ds_read_b128 lane group then sit on distinct banks (also at row offsets
    // 1 and 2, i.e. for all three taps).
    const int l8 = lane >> 3, lc = lane & 7;
    unsigned wbase[PWW];
    // pixel pieces of this wave: piece index wid + j*NW (interleaved: the waves' counts differ by at most one);
    // LDS row i of the slot = padded position q0 - 1 + i
    const int npieces = (BM + 2 + 7) >> 3;
    const int cntx = __builtin_amdgcn_readfirstlane((L16 && !loader) ? 0 : (wid < npieces ? (npieces - 1 - wid) / NWD + 1 : 0));   // (a consumer of the 12- / 16-wave forms issues nothing)
    const int pxa = (cntx + 1) >> 1, pxb = cntx >> 1;      // issued with tap 0 / tap 1 of an earlier stage
    int xbase[PXW], xok[PXW];
    const int rowpitch = a.W * rowbytes;
    auto setup_tile = [&](int gi) __attribute__((always_inline)) {                        // DMA source offsets of tile gi
        n0 = (gi % nt) * BN;
        q0 = (gi / nt) * BM;                               // first padded position of this tile
#pragma unroll
        for (int j = 0; j < PWW; ++j) {
            const int row = (wid * PWW + j) * 8 + l8;
            wbase[j] = (unsigned)(n0 + row) * (unsigned)(9 * rowbytes) + (unsigned)((lc ^ ((row >> 1) & 7)) * 16);
        }
#pragma unroll
        for (int j = 0; j < PXW; ++j) {
            const int i = (wid + j * NWD) * 8 + l8;
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
    f32x16 acc[TN][TMMAX];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TMMAX; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // ---- read side.  k-step ks, lane half h reads source chunk 2 ks + h of its row, stored at position (2 ks + h) ^ key(row):
    // byte offset ((h ^ key) << 4) ^ (ks << 5), key = (row >> 1) & 7 (weights: row r; pixels: row r + kj).  Computed, not
    // tabulated: a table indexed by the tap would live in scratch memory, whose loads count against vmcnt like the DMA.
    const int swa0 = (h ^ ((r >> 1) & 7)) << 4;
    const int rdA = (wn * TN * 32 + r) * 128;
    const int rdX = (pt0 * 32 + r) * 128;
    const int tapstep = a.flip ? -rowbytes : rowbytes;

    // ---- epilogue of one tile (as k_conv_igemm): v = acc + shift + res ; relu ; v *= (mask > 0) ; 8 consecutive channels per access
    // chain mode: wait until the tiles of the previous layer that tile (layer, q0) reads are complete.  Wave 0 polls, one counter
    // per lane; the others wait at the barrier it joins afterwards.
    bool gave_up = false;
#ifdef RS_WSTAMP
    unsigned ws_arr = 0, ws_rel = 0, ws_misc = 0;
    int ws_tile = 0;
    RS_WS(ws_misc, 0);
#endif
#ifdef RS_STAMP
    long long stamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int item = 0;
#endif
    auto chain_wait = [&]() __attribute__((always_inline)) {
        if constexpr (CHAIN) {
            if (layer == 0) return;
            if (wid == 0) {
                if (!gave_up) {
                    const int lo = max(q0 - 1 - Wp, 0) / BM;
                    const int hi = min(a.mtiles - 1, (q0 + BM + Wp) / BM);
                    const bool need = lo + lane <= hi;
                    const int *p = arg.ws + 4 + (layer - 1) * a.mtiles + (need ? lo + lane : lo);
                    int spins = 0;
                    for (;;) {
                        const int v = need ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : nt;
                        if (__builtin_amdgcn_ballot_w64(v < nt) == 0) break;          // (a ballot: uniform for the compiler too)
                        if (++spins > arg.spin_limit) {            // give up: wrong results instead of a hung GPU; the host is told
                            gave_up = true;
                            if (lane == 0) __hip_atomic_store(arg.ws + 1, 0x40000000 | (layer << 16) | (q0 / BM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                }
                if (arg.acquire) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // (no instruction: keeps the loads below the poll)
                }
            }
            __builtin_amdgcn_s_barrier();
        }
    };
    auto gldsX = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned dst) __attribute__((always_inline)) {
        if constexpr (CHAIN) glds16_sc1(rsrc, voff, dst); else glds16(rsrc, voff, dst);
    };
    auto store_tile = [&](int q0c, int n0c) __attribute__((always_inline)) {
        // exactly cnt x TN x 2 store instructions per wave, whatever the tile (see gst16)
        if (DCF_DBG(a) & 4) {
#pragma unroll
            for (int j = 0; j < TMMAX; ++j)
                if (j < cnt)
#pragma unroll
                    for (int k = 0; k < TN * 2; ++k) gst16_dummy(dstY);
            return;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TMMAX; ++j) acc_rows8(acc[i][j]);
        // Two rounds of independent loads instead of one dependent load per store (behind an asm store -- a compiler barrier --
        // each would pay its own latency): every residual vector of the wave's tiles, folded into the accumulators with the
        // shift and the ReLU; then every mask vector; then the stores.
        int mrow[TMMAX];
        bool valid[TMMAX];
#pragma unroll
        for (int j = 0; j < TMMAX; ++j) {
            const int p = q0c + (pt0 + j) * 32 + r;
            const int R = p / Wp, c = p - R * Wp;
            valid[j] = (j < cnt) && !(p >= a.Q || c < 1 || c > a.W);      // padding position: no output
            mrow[j] = R * a.W + c - 1;
        }
        auto voff = [&](int j, int i, int pp) { return (size_t)mrow[j] * a.Cn + n0c + (wn * TN + i) * 32 + 16 * pp + 8 * h; };
        if (res) {
            uint4 rr[TMMAX][TN][2];
#pragma unroll
            for (int j = 0; j < TMMAX; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        if constexpr (CHAIN) {
                            // (sc1: the residual of a chain layer is an earlier layer's output, written in this launch)
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srcR, valid[j] ? (int)(voff(j, i, pp) * sizeof(T)) : (int)OOB, 0, 16);
                            rr[j][i][pp] = make_uint4(v[0], v[1], v[2], v[3]);
                        } else
                            rr[j][i][pp] = valid[j] ? *reinterpret_cast<const uint4 *>(res + voff(j, i, pp)) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TMMAX; ++j)
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
        if (lshift) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    float sh[8];
                    ld8(lshift + n0c + (wn * TN + i) * 32 + 16 * pp + 8 * h, sh);
#pragma unroll
                    for (int j = 0; j < TMMAX; ++j)
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[i][j][8 * pp + k] += sh[k];
                }
        }
        if (lrelu) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TMMAX; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[i][j][k] = fmaxf(acc[i][j][k], 0.f);
        }
        if (mask) {
            uint4 mm[TMMAX][TN][2];
#pragma unroll
            for (int j = 0; j < TMMAX; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        mm[j][i][pp] = valid[j] ? *reinterpret_cast<const uint4 *>(mask + voff(j, i, pp)) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TMMAX; ++j)
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
        for (int j = 0; j < TMMAX; ++j) {
            if (j >= cnt) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = acc[i][j][8 * pp + k];
                    gst16<CHAIN ? 16 : 0>(dstY, valid[j] ? (unsigned)(voff(j, i, pp) * sizeof(T)) : OOB, pack8<T>(v));
                }
        }
    };

    // Main loop, specialised on the wave's tile count C and on its pixel-piece count CX (both wave-uniform: one dispatch, no
    // branches between the MFMAs, every wait an immediate).  Step t = 3 s + kj issues "group t": the weights of tap t + DW
    // (ring slot (t + DW) % NSW) and, on a stage's taps 0 / 1, the first / second half of this wave's pieces of the pixel
    // tile of stage s + DX.  Past the end of the K loop the same instructions are issued with out-of-range offsets (they
    // write zeros into slots nobody reads any more), so every group has the same size and the number of instructions issued
    // after the group a step depends on is a compile-time constant:
    //     step t needs group t - DW (its weights) and, on tap 0, groups 3 (s - DX) and 3 (s - DX) + 1 (its pixel tile).
    // Per step: counted wait -> barrier (everyone's pieces have landed, everyone is done with the slots about to be
    // refilled) -> issue group t -> MFMAs.
    auto main_loop = [&](auto CNT, auto CNTX, auto NOISSUE) {
        constexpr int C = decltype(CNT)::value, CX = decltype(CNTX)::value;
        constexpr bool NI = decltype(NOISSUE)::value;          // 16-wave form, consumer: this wave issues no DMA piece at all
        constexpr int PXA = (CX + 1) / 2, PXB = CX / 2;
        constexpr int DS = (2 + DW) / 3 > DX ? (2 + DW) / 3 : DX;          // stages the bookkeeping looks ahead
        constexpr int A0 = rs_allowed(0, DW, DX, PWW, PXA, PXB), A1 = rs_allowed(1, DW, DX, PWW, PXA, PXB), A2 = rs_allowed(2, DW, DX, PWW, PXA, PXB);
        static_assert(A0 < 64 && A1 < 64 && A2 < 64, "vmcnt range");
        constexpr int AS = (DX - 1) * (3 * PWW + CX);                      // S3: pieces issued after the stage's own group
        // The previous tile's stores (the first tile: as many dummy ones) are issued after the tile's first groups and are left in
        // flight: a step that depends on one of those first groups may leave them outstanding too.
        constexpr int ST = C * TN * 2;
        constexpr int BK0 = (3 * DX - 1 < DW) ? 3 * DX - 1 : DW;            // groups back a tap-0 step depends on (rs_allowed)
        static_assert(AS + ST < 64 && A0 + ST < 64 && A1 + ST < 64 && A2 + ST < 64, "vmcnt range");
        // stage coordinates of stages s .. s + DS: weight offset of the stage's tap 0, pixel offset, kernel row (3 = past the end)
        int wst[DS + 1], xst[DS + 1], kis[DS + 1];
        int lki = 0, lcc = 0;
        auto stage_entry = [&](int d) __attribute__((always_inline)) {
            kis[d] = lki > 2 ? 3 : lki;
            wst[d] = (a.flip ? (2 - lki) * 3 + 2 : lki * 3) * rowbytes + lcc * 128;
            xst[d] = (lki - 1) * rowpitch + lcc * 128;
            if (++lcc == cchunks) { lcc = 0; ++lki; }
        };
        auto issue_w = [&](int d, int kj, int slot) __attribute__((always_inline)) {                  // weights of tap kj of stage s + d
            if constexpr (NI) return;
            const unsigned dst = __builtin_amdgcn_readfirstlane(ldsW0 + slot * WSLOT + wid * PWW * 1024);
            const bool ok = kis[d] < 3 && !(DCF_DBG(a) & 16);
            const unsigned koff = (unsigned)(wst[d] + kj * tapstep);
#pragma unroll
            for (int j = 0; j < PWW; ++j) glds16(srcW, ok ? wbase[j] + koff : OOB, dst + j * 1024);
        };
        auto issue_x = [&](int d, int slot, int j0, int j1) __attribute__((always_inline)) {          // pieces j0 .. j1-1 of the pixel tile of stage s + d
            if constexpr (NI) return;
            const int ki = kis[d];
#pragma unroll
            for (int j = 0; j < PXW; ++j)
                if (j >= j0 && j < j1) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(ldsX0 + slot * XSLOT + (wid + j * NWD) * 1024);
                    gldsX(srcX, (((xok[j] >> ki) & 1) && !(DCF_DBG(a) & 2)) ? (unsigned)(xbase[j] + xst[d]) : OOB, dst);
                }
        };
        // stage bookkeeping of a fresh tile + its first groups.  WPART / XPART: issue the weight / the pixel pieces of those groups
        // (both: the plain kernel's prologue; chain mode issues the weights of the next item before it stores the current one --
        // they depend on nothing -- and the pixels once the tiles they come from have been published)
        auto begin_tile_parts = [&](auto WPART, auto XPART) __attribute__((always_inline)) {
            constexpr bool DOW = decltype(WPART)::value, DOX = decltype(XPART)::value;
            if constexpr (DOW) {
                lki = 0; lcc = 0;
#pragma unroll
                for (int d = 0; d <= DS; ++d) stage_entry(d);
            }
        // prologue: groups -3 DX .. -1, then everything landed (the first taps need their data at once anyway)
            {
                int wsl = 0, xsl = 0;
                if constexpr (S3) {
#pragma unroll
                    for (int sv = 0; sv < DX; ++sv) {
                        if constexpr (DOX) issue_x(sv, sv, 0, CX);
                        if constexpr (DOW)
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw) issue_w(sv, kw, sv * 3 + kw);
                    }
                } else
#pragma unroll
                for (int u = -3 * DX; u < 0; ++u) {
                    const int v = u + 3 * DX, sv = v / 3, kj = v - 3 * sv;                 // pixel tile of stage sv < DX
                    if constexpr (DOX) {
                        if (kj == 0) issue_x(sv, xsl, 0, PXA);
                        if (kj == 1) issue_x(sv, xsl, PXA, CX);
                    }
                    if (kj == 1) ++xsl;
                    if (u + DW >= 0) { if constexpr (DOW) issue_w((u + DW) / 3, (u + DW) % 3, wsl); ++wsl; }
                }
            }
        };
        auto begin_tile = [&]() __attribute__((always_inline)) { begin_tile_parts(std::true_type(), std::true_type()); };
        RS_T(0); RS_T(1);
        begin_tile();
        // The previous tile's ST stores sit between a tile's first groups and its loop in the vmcnt queue, and the first steps' waits
        // allow for them.  The FIRST tile has none: its waits use the plain immediates (exact without stores).  (Rounds 2-5 issued
        // ST out-of-range stand-in stores here instead: 4-10 store instructions per wave in front of the first tap of every launch --
        // -DRS_WSTAMP: 3.3 k cycles from kernel entry to "first groups issued" on the 192-channel stage, 9.5 k on the 128-channel one.)
        bool stores_pending = false;
        RS_T(2);
#ifdef RS_WSTAMP
        RS_WS(ws_misc, 1);
#endif
        for (;;) {
        if constexpr (CHAIN) make_descs();
        int wsr = 0, wsi = DW % NSW, xsr = 0, xsi = DX % NSX;         // ring slots: read / issue
        if constexpr (PF) {
            constexpr int CC = C > 0 ? C : 1;
            uint4 fa0[TN], fb0[CC], fa1[TN], fb1[CC];
            auto load_frags = [&](uint4 (&fa)[TN], uint4 (&fb)[CC], int kj, int ks) __attribute__((always_inline)) {
#if defined(RS_ABL) && (RS_ABL & 2)
                if (kj + ks != 0) return;                  // timing ablation (variant builds): fragments read once per stage only
#endif
                const char *pw = lds + wsr * WSLOT + rdA;
                const char *px = lds + NSW * WSLOT + xsr * XSLOT + rdX + kj * 128;
                const int swx0 = (h ^ (((r + kj) >> 1) & 7)) << 4;
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4 *>(pw + i * 32 * 128 + (swa0 ^ (ks << 5)));
#pragma unroll
                for (int j = 0; j < C; ++j) fb[j] = *reinterpret_cast<const uint4 *>(px + j * 32 * 128 + (swx0 ^ (ks << 5)));
            };
            auto mfmas = [&](const uint4 (&fa)[TN], const uint4 (&fb)[CC]) __attribute__((always_inline)) {
#if defined(RS_ABL) && (RS_ABL & 4)
                return;
#endif
#pragma unroll
                for (int j = 0; j < C; ++j)
#pragma unroll
                    for (int i = 0; i < TN; ++i) Mma<T>::run(fa[i], fb[j], acc[i][j]);
            };
            auto wait_tap = [&](int kj, int t) __attribute__((always_inline)) {          // the synchronisation in front of tap t
                if constexpr (S3) {                        // stage-granular form: one wait + barrier in front of a stage's first tap
                    if (kj != 0) return;
                    const int sn = t / 3;
                    if (sn < DX && stores_pending) wait_vmcnt<AS + ST>(); else wait_vmcnt<AS>();
                    wait_lgkm0();
#ifdef RS_WSTAMP
                    if (ws_tile == 0 && sn < 64) RS_WS(ws_arr, sn);
#endif
                    __builtin_amdgcn_s_barrier();
#ifdef RS_WSTAMP
                    if (ws_tile == 0 && sn < 64) RS_WS(ws_rel, sn);
#endif
                    return;
                }
                const bool early = stores_pending && t < (kj == 0 ? BK0 : DW);
                if (kj == 0) { if (early) wait_vmcnt<A0 + ST>(); else wait_vmcnt<A0>(); }
                else if (kj == 1) { if (early) wait_vmcnt<A1 + ST>(); else wait_vmcnt<A1>(); }
                else { if (early) wait_vmcnt<A2 + ST>(); else wait_vmcnt<A2>(); }
                wait_lgkm0();
#ifdef RS_WSTAMP
                if (ws_tile == 0 && t < 64) RS_WS(ws_arr, t);
#endif
#if !(defined(RS_ABL) && (RS_ABL & 8))
                __builtin_amdgcn_s_barrier();
#endif
#ifdef RS_WSTAMP
                if (ws_tile == 0 && t < 64) RS_WS(ws_rel, t);
#endif
            };
            auto issue_quarter = [&](int kj, int part) __attribute__((always_inline)) {   // quarter `part` of group t (tap kj of the current stage)
                if constexpr (NI) return;
#if defined(RS_ABL) && (RS_ABL & 1)
                return;                                    // timing ablation (variant builds): no DMA piece inside the loop
#endif
                const int xj0 = kj == 0 ? 0 : PXA, xj1 = kj == 0 ? PXA : (kj == 1 ? CX : PXA);
                const int dw = (kj + DW) / 3, kw = (kj + DW) % 3;
                const unsigned dstw = __builtin_amdgcn_readfirstlane(ldsW0 + wsi * WSLOT + wid * PWW * 1024);
                const bool okw = kis[dw] < 3 && !(DCF_DBG(a) & 16);
                const unsigned koff = (unsigned)(wst[dw] + kw * tapstep);
#pragma unroll
                for (int j = 0; j < PWW; ++j)
                    if ((j & 3) == part) glds16(srcW, okw ? wbase[j] + koff : OOB, dstw + j * 1024);
                const int kix = kis[DX];
#pragma unroll
                for (int j = 0; j < PXW; ++j)
                    if (j >= xj0 && j < xj1 && ((PWW + j - xj0) & 3) == part) {
                        const unsigned dst = __builtin_amdgcn_readfirstlane(ldsX0 + xsi * XSLOT + (wid + j * NWD) * 1024);
                        gldsX(srcX, (((xok[j] >> kix) & 1) && !(DCF_DBG(a) & 2)) ? (unsigned)(xbase[j] + xst[DX]) : OOB, dst);
                    }
            };
            auto issue_group = [&](int kj) __attribute__((always_inline)) {               // a wave without tiles: the whole group at once
#if defined(RS_ABL) && (RS_ABL & 1)
                return;
#endif
                if constexpr (S3) {                        // the stage's group, behind the stage's barrier
                    if (kj != 0) return;
                    issue_x(DX, xsi, 0, CX);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) issue_w(DX, kw, wsi + kw);
                    return;
                }
                issue_w((kj + DW) / 3, (kj + DW) % 3, wsi);
                if (kj == 0) issue_x(DX, xsi, 0, PXA);
                if (kj == 1) issue_x(DX, xsi, PXA, CX);
            };
            wait_tap(0, 0);
            if constexpr (C > 0) { load_frags(fa0, fb0, 0, 0); issue_quarter(0, 0); } else issue_group(0);
            for (int s = 0; s < nstage; ++s) {
#pragma unroll
                for (int kj = 0; kj < 3; ++kj) {
                    const int kjn = kj == 2 ? 0 : kj + 1;
                    if constexpr (C > 0) {
                        load_frags(fa1, fb1, kj, 1);
                        __builtin_amdgcn_sched_barrier(0);
                        mfmas(fa0, fb0);
                        __builtin_amdgcn_sched_barrier(0);
                        issue_quarter(kj, 1);
                        load_frags(fa0, fb0, kj, 2);
                        __builtin_amdgcn_sched_barrier(0);
                        mfmas(fa1, fb1);
                        __builtin_amdgcn_sched_barrier(0);
                        issue_quarter(kj, 2);
                        load_frags(fa1, fb1, kj, 3);
                        __builtin_amdgcn_sched_barrier(0);
                        mfmas(fa0, fb0);
                        __builtin_amdgcn_sched_barrier(0);
                        issue_quarter(kj, 3);
                    }
                    // the state of tap t + 1 ...
                    wsr = wsr + 1 == NSW ? 0 : wsr + 1;
                    wsi = wsi + 1 == NSW ? 0 : wsi + 1;
                    if (kj == 2) {
                        xsr = xsr + 1 == NSX ? 0 : xsr + 1;
                        xsi = xsi + 1 == NSX ? 0 : xsi + 1;
#pragma unroll
                        for (int d = 0; d < DS; ++d) { wst[d] = wst[d + 1]; xst[d] = xst[d + 1]; kis[d] = kis[d + 1]; }
                        stage_entry(DS);
                    }
                    // ... its synchronisation and its first fragments, under this tap's last k-step
                    const bool has_next = !(kj == 2 && s == nstage - 1);
                    if (has_next) {
                        wait_tap(kjn, 3 * s + kj + 1);
                        if constexpr (C > 0) load_frags(fa0, fb0, kjn, 0);
                    }
                    if constexpr (C > 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        mfmas(fa1, fb1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (has_next) {
                        if constexpr (C > 0) issue_quarter(kjn, 0); else issue_group(kjn);
                    }
                }
            }
        } else
        for (int s = 0; s < nstage; ++s) {
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) {
                if constexpr (S3) {
                    if (kj == 0) {
                        if (s < DX && stores_pending) wait_vmcnt<AS + ST>(); else wait_vmcnt<AS>();
                        wait_lgkm0();             // (see below)
#ifdef RS_WSTAMP
                        if (ws_tile == 0 && s < 64) RS_WS(ws_arr, s);
#endif
                        __builtin_amdgcn_s_barrier();
#ifdef RS_WSTAMP
                        if (ws_tile == 0 && s < 64) RS_WS(ws_rel, s);
#endif
#ifdef RS_STAMP
                        if (s == 0) RS_T(3);
#endif
                    }
                } else {
                    const bool early = stores_pending && 3 * s + kj < (kj == 0 ? BK0 : DW);       // the group this step needs went out before the stores
                    if (kj == 0) { if (early) wait_vmcnt<A0 + ST>(); else wait_vmcnt<A0>(); }
                    else if (kj == 1) { if (early) wait_vmcnt<A1 + ST>(); else wait_vmcnt<A1>(); }
                    else { if (early) wait_vmcnt<A2 + ST>(); else wait_vmcnt<A2>(); }
                    // Behind this barrier other waves aim DMA pieces at the slots the previous tap was read from: every fragment
                    // read of this wave must have RETURNED by now.  The compiler may sink the previous tap's last MFMA -- and the
                    // lgkmcnt wait in front of it -- below a raw s_barrier (tools/audit_barrier_lds.py found 885 such barriers in
                    // the round-5 object: a race the L2 latency won); the explicit wait makes it a dependency.
                    wait_lgkm0();
#ifdef RS_WSTAMP
                    if (ws_tile == 0 && 3 * s + kj < 64) RS_WS(ws_arr, 3 * s + kj);
#endif
                    __builtin_amdgcn_s_barrier();
#ifdef RS_WSTAMP
                    if (ws_tile == 0 && 3 * s + kj < 64) RS_WS(ws_rel, 3 * s + kj);
#endif
#ifdef RS_STAMP
                    if (s == 0 && kj == 0) RS_T(3);
#endif
                }
                // (issuing the second half-workgroup's DMA after its MFMAs instead -- waves w and w + 4 share a SIMD -- was
                // measured: no gain, 26.7 -> 27.3 us on the 128-channel stage)
                // The group's pieces go out one per K-quarter, between the MFMAs (DCF_RS_SPREAD): a burst of every wave's
                // pieces right after the barrier holds the waves at their issue (~60-180 cycles per piece) while the
                // matrix pipes idle.  Waves without tiles (C == 0) issue theirs at once.
                const int xj0 = kj == 0 ? 0 : PXA, xj1 = kj == 0 ? PXA : (kj == 1 ? CX : PXA);       // kj == 2: none
                auto issue_part = [&](int part) {
                    if constexpr (NI) return;
                    if constexpr (S3) {                    // the stage's group (3 PWW weight + CX pixel pieces) over its 12 K-quarters
                        const int q12 = kj * 4 + part;
                        const bool okw = kis[DX] < 3 && !(DCF_DBG(a) & 16);
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                            for (int j = 0; j < PWW; ++j)
                                if ((kw * PWW + j) % 12 == q12) {
                                    const unsigned dstw = __builtin_amdgcn_readfirstlane(ldsW0 + (wsi - kj + kw) * WSLOT + (wid * PWW + j) * 1024);
                                    glds16(srcW, okw ? wbase[j] + (unsigned)(wst[DX] + kw * tapstep) : OOB, dstw);
                                }
                        const int kix = kis[DX];
#pragma unroll
                        for (int j = 0; j < PXW; ++j)
                            if (j < CX && (3 * PWW + j) % 12 == q12) {
                                const unsigned dst = __builtin_amdgcn_readfirstlane(ldsX0 + xsi * XSLOT + (wid + j * NWD) * 1024);
                                gldsX(srcX, (((xok[j] >> kix) & 1) && !(DCF_DBG(a) & 2)) ? (unsigned)(xbase[j] + xst[DX]) : OOB, dst);
                            }
                        return;
                    }
                    const int dw = (kj + DW) / 3, kw = (kj + DW) % 3;
                    const unsigned dstw = __builtin_amdgcn_readfirstlane(ldsW0 + wsi * WSLOT + wid * PWW * 1024);
                    const bool okw = kis[dw] < 3 && !(DCF_DBG(a) & 16);
                    const unsigned koff = (unsigned)(wst[dw] + kw * tapstep);
#pragma unroll
                    for (int j = 0; j < PWW; ++j)
                        if ((j & 3) == part) glds16(srcW, okw ? wbase[j] + koff : OOB, dstw + j * 1024);
                    const int kix = kis[DX];
#pragma unroll
                    for (int j = 0; j < PXW; ++j)
                        if (j >= xj0 && j < xj1 && ((PWW + j - xj0) & 3) == part) {
                            const unsigned dst = __builtin_amdgcn_readfirstlane(ldsX0 + xsi * XSLOT + (wid + j * NWD) * 1024);
                            gldsX(srcX, (((xok[j] >> kix) & 1) && !(DCF_DBG(a) & 2)) ? (unsigned)(xbase[j] + xst[DX]) : OOB, dst);
                        }
                };
                if (!DCF_RS_SPREAD || C == 0 || (DCF_DBG(a) & 1)) {
                    if constexpr (S3) {
                        if (kj == 0) {
                            issue_x(DX, xsi, 0, CX);
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw) issue_w(DX, kw, wsi + kw);
                        }
                    } else {
                        issue_w((kj + DW) / 3, (kj + DW) % 3, wsi);
                        if (kj == 0) issue_x(DX, xsi, 0, PXA);
                        if (kj == 1) issue_x(DX, xsi, PXA, CX);
                    }
                }
                if constexpr (C > 0) if (!(DCF_DBG(a) & 1)) {
                    const char *pw = lds + wsr * WSLOT + rdA;
                    const char *px = lds + NSW * WSLOT + xsr * XSLOT + rdX + kj * 128;
                    const int swx0 = (h ^ (((r + kj) >> 1) & 7)) << 4;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        uint4 fa[TN], fb[C];
#pragma unroll
                        for (int i = 0; i < TN; ++i) fa[i] = *reinterpret_cast<const uint4 *>(pw + i * 32 * 128 + (swa0 ^ (ks << 5)));
#pragma unroll
                        for (int j = 0; j < C; ++j) fb[j] = *reinterpret_cast<const uint4 *>(px + j * 32 * 128 + (swx0 ^ (ks << 5)));
#pragma unroll
                        for (int j = 0; j < C; ++j)
#pragma unroll
                            for (int i = 0; i < TN; ++i) Mma<T>::run(fa[i], fb[j], acc[i][j]);
                        if (DCF_RS_SPREAD) issue_part(ks);
                    }
                }
                wsr = wsr + 1 == NSW ? 0 : wsr + 1;
                wsi = wsi + 1 == NSW ? 0 : wsi + 1;
            }
            xsr = xsr + 1 == NSX ? 0 : xsr + 1;
            xsi = xsi + 1 == NSX ? 0 : xsi + 1;
#pragma unroll
            for (int d = 0; d < DS; ++d) { wst[d] = wst[d + 1]; xst[d] = xst[d + 1]; kis[d] = kis[d + 1]; }
            stage_entry(DS);
        }
        RS_T(4);
#ifdef RS_WSTAMP
        if (ws_tile == 0) RS_WS(ws_misc, 2);
#endif
        wait_vmcnt<0>();              // the trailing dummy pieces still target this workgroup's LDS
        RS_T(5);
#ifdef RS_WSTAMP
        if (ws_tile == 0) RS_WS(ws_misc, 3);
#endif
        const int q0c = q0, n0c = n0;
        if constexpr (CHAIN) {
            // Publish this tile, then move on to the next work item -- (layer, next tile of this workgroup) or (layer + 1, its
            // first tile) -- once the tiles that one reads have been published.  The next item's first WEIGHT groups go out
            // before this tile's epilogue (they depend on nothing and land under the epilogue, the store drain and the wait);
            // its pixels may not exist yet and follow the wait.
            int ngidx = gidx + wpx, nlayer = layer;
            if (ngidx >= chunk_hi) { nlayer = __builtin_amdgcn_readfirstlane(layer + 1); ngidx = gidx_first; }
            const bool more = nlayer < arg.nlayers;
            if (more) {
                wait_lgkm0();
                __builtin_amdgcn_s_barrier();                          // every wave is done with the rings
                if (nlayer != layer) bind_weights(nlayer);
                if (ngidx != gidx) setup_tile(ngidx);                  // (one tile per layer: the DMA offsets of the tile stay)
                begin_tile_parts(std::true_type(), std::false_type());
            }
            store_tile(q0c, n0c);
            RS_T(6);
            wait_vmcnt<0>();                                           // EVERY storing wave drains its write-through stores ...
            RS_T(7);
            __builtin_amdgcn_s_barrier();                              // ... before one lane signals for all (and the rings are idle)
            if (tid == 0) __hip_atomic_fetch_add(arg.ws + 4 + layer * a.mtiles + q0c / BM, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            RS_T(8);
#ifdef RS_STAMP
            if (tid == 0 && blockIdx.x < RS_STAMP_WGS && item < RS_STAMP_ITEMS) {
#pragma unroll
                for (int k9 = 0; k9 < 9; ++k9) g_rs_stamps[blockIdx.x][item][k9] = stamp[k9];
                g_rs_stamps[blockIdx.x][item][9] = ((long long)layer << 32) | (unsigned)(q0c / BM * nt + n0c / BN);
            }
            ++item;
#endif
            if (!more) break;
            gidx = ngidx;
            if (nlayer != layer) { layer = nlayer; bind_layer(layer); }
            RS_T(0);
            chain_wait();
            RS_T(1);
            begin_tile_parts(std::false_type(), std::true_type());
            RS_T(2);
            // Everything of the first groups has to have landed before the loop's counted waits take over: they price the
            // prologue's groups in the plain kernel's issue order (weights and pixels interleaved, then the stores), which this
            // split order is not; with nothing outstanding here every such wait is trivially met, and the waits of the later
            // steps count main-loop groups only.
            wait_vmcnt<0>();
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TMMAX; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
            continue;
        }
        gidx += wpx;
        const bool more = gidx < chunk_hi;
        if (more) {
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();                              // every wave is done with the rings
            setup_tile(gidx);
            begin_tile();                                              // in flight under the epilogue below
        }
        store_tile(q0c, n0c);
        stores_pending = true;
#ifdef RS_WSTAMP
        if constexpr (!CHAIN) {
            if (ws_tile == 0) {
                RS_WS(ws_misc, 4);
                if (!more) { wait_vmcnt<0>(); RS_WS(ws_misc, 5); }
            }
            ++ws_tile;
        }
#endif
        if (!more) break;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TMMAX; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        }
    };
    // dispatch on (tiles, pixel pieces) of this wave; the host's plan keeps both inside the instantiated ranges
    if constexpr (L16) {
        // consumers: (tiles, no pieces, no issue); loaders: (no tiles, their pieces)
        if (!loader) {
            switch (cnt) {
            case 0: main_loop(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), std::true_type()); break;
            case 1: main_loop(std::integral_constant<int, 1>(), std::integral_constant<int, 0>(), std::true_type()); break;
            case 2: main_loop(std::integral_constant<int, (TMMAX >= 2 ? 2 : TMMAX)>(), std::integral_constant<int, 0>(), std::true_type()); break;
            case 3: main_loop(std::integral_constant<int, (TMMAX >= 3 ? 3 : TMMAX)>(), std::integral_constant<int, 0>(), std::true_type()); break;
            case 4: main_loop(std::integral_constant<int, (TMMAX >= 4 ? 4 : TMMAX)>(), std::integral_constant<int, 0>(), std::true_type()); break;
            default: main_loop(std::integral_constant<int, TMMAX>(), std::integral_constant<int, 0>(), std::true_type()); break;
            }
        } else {
#define DCF_RS_LX(X_) case X_: main_loop(std::integral_constant<int, 0>(), std::integral_constant<int, (PXW >= X_ ? X_ : PXW)>(), std::false_type()); break;
            switch (cntx) {
                DCF_RS_LX(0) DCF_RS_LX(1) DCF_RS_LX(2) DCF_RS_LX(3) DCF_RS_LX(4) DCF_RS_LX(5) DCF_RS_LX(6) DCF_RS_LX(7)
                DCF_RS_LX(8) DCF_RS_LX(9) DCF_RS_LX(10) DCF_RS_LX(11) DCF_RS_LX(12)
            default: main_loop(std::integral_constant<int, 0>(), std::integral_constant<int, PXW>(), std::false_type()); break;
            }
#undef DCF_RS_LX
        }
    } else {
#define DCF_RS_CX(C_)                                                                                               \
    switch (cntx) {                                                                                                  \
    case 0: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, 0>(), std::false_type()); break;                   \
    case 1: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 1 ? 1 : PXW)>(), std::false_type()); break; \
    case 2: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 2 ? 2 : PXW)>(), std::false_type()); break; \
    case 3: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 3 ? 3 : PXW)>(), std::false_type()); break; \
    case 4: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 4 ? 4 : PXW)>(), std::false_type()); break; \
    case 5: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 5 ? 5 : PXW)>(), std::false_type()); break; \
    case 6: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, (PXW >= 6 ? 6 : PXW)>(), std::false_type()); break; \
    default: main_loop(std::integral_constant<int, C_>(), std::integral_constant<int, PXW>(), std::false_type()); break;                \
    }
    switch (cnt) {
    case 0: DCF_RS_CX(0) break;
    case 1: DCF_RS_CX(1) break;
    case 2: DCF_RS_CX((TMMAX >= 2 ? 2 : TMMAX)) break;
    case 3: DCF_RS_CX((TMMAX >= 3 ? 3 : TMMAX)) break;
    case 4: DCF_RS_CX((TMMAX >= 4 ? 4 : TMMAX)) break;
    default: DCF_RS_CX(TMMAX) break;
    }
#undef DCF_RS_CX
    }
#ifdef RS_WSTAMP
    if constexpr (!CHAIN) {
        if (blockIdx.x < RS_WS_WGS && wid16 < 8) {
            g_rs_wstamps[blockIdx.x][wid16][0][lane] = ws_arr;
            g_rs_wstamps[blockIdx.x][wid16][1][lane] = ws_rel;
            g_rs_wstamps[blockIdx.x][wid16][2][lane] = ws_misc;
        }
    }
#endif
    chain_finish();
}

// Tile shape of a launch.  kind 0: 128 channels x up to 320 positions (waves 4 x 2, up to 5 position tiles per wave);
// kind 1: 64 channels x up to 384 positions (waves 2 x 4, up to 3 per wave); kind 2: 64 channels x up to 128 positions
// (waves 2 x 4, one tile per wave) with the DMA running 5 taps / 2 stages ahead -- the small-M layers, whose steps are
// too short to hide the L2 latency behind two taps.  npt is picked so that the tile count fills the CUs in whole rounds
// and the waves' shares are even.  (DCF_RS_KIND / DCF_RS_NPT force a choice: experiments.)
struct RsPlan { int kind, npt; };
struct RsKind { int BN, WM, TMMAX, ahead, nthreads, per_cu; };
// (4-wave workgroups, two per CU with <= 80 KB of LDS each so that one's prologue and epilogue overlap the other's taps, were
// measured as kinds {64, 2, 2, 2, 256, 2} and {128, 2, 2, 1, 256, 2}: never ahead of these three, removed)
static const RsKind RS_KINDS[3] = {{128, 2, 5, 2, 512, 1}, {64, 4, 3, 2, 512, 1}, {64, 4, 1, 5, 512, 1}};

static RsPlan rs_plan(int64_t Q, int Cn)
{
    static DcfOpt ek_o("RS_KIND"), en_o("RS_NPT");
    const char *ek = ek_o.str(), *en = en_o.str();
    const int ncu = 256;
    RsPlan best = {-1, 0};
    double best_t = 1e30;
    for (int kind = 0; kind < 3; ++kind) {
        const RsKind &k = RS_KINDS[kind];
        if (Cn % k.BN) continue;
        if (ek && atoi(ek) != kind) continue;
        for (int npt = 1; npt <= k.WM * k.TMMAX; ++npt) {
            if (en && atoi(en) != npt) continue;
            const int64_t tiles = (Q + 32 * npt - 1) / (32 * npt) * (Cn / k.BN);
            const int64_t rounds = (tiles + ncu * k.per_cu - 1) / (ncu * k.per_cu);
            const int per_wave = (npt + k.WM - 1) / k.WM;
            // cycles per tap step on a CU: MFMAs of the busiest SIMD (2 waves), the L2 -> LDS transfer at ~28 B/clk, a fixed
            // cost of the wait + barrier + issue sequence, and the L2 latency spread over the taps the DMA runs ahead
            const double mfma = 2.0 * 4 * per_wave * 32;
            const double dma = ((32.0 * npt + 2) / 3.0 + k.BN) * 128.0 / 28.0;
            const double step = std::max(std::max(mfma, dma) + 220.0, 1800.0 / k.ahead);
            const double t = rounds * (step + 40.0 * per_wave /* epilogue share */);
            if (t < best_t) { best_t = t; best = {kind, npt}; }
        }
    }
    return best;
}

}  // namespace
