// conv_chain.hip -- chain launches of the row-sharing 3x3 / stride-1 kernel (conv_rs_kernel.h, RsChainArgs): the bodies of a
// residual stage (/root/reference/model.py:32-41, :48-60) in ONE persistent launch, forward or input gradient.
#include "conv_rs_kernel.h"

// ---------------------------------------------------------------- chain launches (round 5)
namespace {
// Does one launch of the chain kernel cover a chain of [B,H,W,C] -> [B,H,W,C] layers?  Every workgroup has to be resident
// at once (a tile waits for its neighbours' previous-layer tiles: conv_rs.hip, RsChainArgs), so the tile count of a
// layer must fit one round of workgroups, and the position tiles a tile's halo touches must fit one wave's poll.
bool rs_chain_plan(int dtype, int B, int H, int W, int C, RsPlan *plan, int *nwg)
{
    if ((dtype != DCF_BF16 && dtype != DCF_F16) || C % 64 || B <= 0 || H <= 0 || W <= 0) return false;
    const int64_t Q = (int64_t)B * H * (W + 2);
    if (Q >= (1ll << 30) || (int64_t)B * H * W * C * 2 >= (1ll << 31)) return false;
    const RsPlan p = rs_plan(Q, C);
    if (p.kind < 0) return false;
    const int BN = RS_KINDS[p.kind].BN, BM = 32 * p.npt;
    const int64_t mtiles = (Q + BM - 1) / BM, nblk = mtiles * (C / BN);
    // every workgroup of a chain launch has to be resident at once: one round of the CUs THIS device (or partition) really has
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    if (nblk > (int64_t)std::min(ncu, 256) * RS_KINDS[p.kind].per_cu) return false;     // more than one round: the plain launches do better (and a chain would starve)
    // More than four channel tiles per position tile: a tile then waits for 3 x 8 or more of a layer's ~90 tiles, which is a
    // grid-wide wait in all but name -- measured on the 512-channel 12x39 camera layers: 21.6 us per layer chained against 19.8
    // as separate launches (profiles/r05e_chain_time.txt); the other cfg2 stages are level or ahead.  Option CHAIN_WIDE=1 lifts it.
    static DcfOpt wide_o("CHAIN_WIDE"); const char *wide = wide_o.str();
    if (C / BN > 4 && !(wide && atoi(wide) == 1)) return false;
    if ((2 * (W + 3) + BM - 1) / BM + 2 > 64) return false;                   // halo tiles polled one per lane
    *plan = p;
    *nwg = (int)((nblk + 7) / 8 * 8);
    return true;
}
}  // namespace

extern "C" int dcf_conv3x3_chain_supported(int dtype, int B, int H, int W, int C, int nlayers)
{
    RsPlan p; int nwg;
    static DcfOpt off_o("CONV_CHAIN"); const char *off = off_o.str();
    if (off && atoi(off) == 0) return 0;
    return nlayers >= 1 && nlayers <= DCF_CHAIN_MAX_LAYERS && rs_chain_plan(dtype, B, H, W, C, &p, &nwg) ? 1 : 0;
}

extern "C" size_t dcf_conv3x3_chain_workspace_bytes(int dtype, int B, int H, int W, int C, int nlayers)
{
    RsPlan p; int nwg;
    if (nlayers < 1 || nlayers > DCF_CHAIN_MAX_LAYERS || !rs_chain_plan(dtype, B, H, W, C, &p, &nwg)) return 0;
    const int64_t Q = (int64_t)B * H * (W + 2);
    const int64_t mtiles = (Q + 32 * p.npt - 1) / (32 * p.npt);
    return (size_t)(4 + (int64_t)nlayers * mtiles) * sizeof(int32_t);
}

extern "C" int dcf_conv3x3_chain(int dtype, const dcf_chain_layer *layers, int nlayers, int B, int H, int W, int C, int flip,
                                 void *ws, dcf_stream_t stream)
{
    RsPlan p; int nwg;
    DCF_REQUIRE(layers && ws, "dcf_conv3x3_chain: null pointer");
    DCF_REQUIRE(nlayers >= 1 && nlayers <= DCF_CHAIN_MAX_LAYERS, "dcf_conv3x3_chain: 1..%d layers (got %d)", DCF_CHAIN_MAX_LAYERS, nlayers);
    DCF_REQUIRE(rs_chain_plan(dtype, B, H, W, C, &p, &nwg), "dcf_conv3x3_chain: unsupported shape / dtype (%d: %dx%dx%d, %d channels) -- ask dcf_conv3x3_chain_supported first", dtype, B, H, W, C);
    static_assert(DCF_CHAIN_MAX_LAYERS == RS_CHAIN_MAX, "header and kernel disagree");
    RsChainArgs ca;
    memset(&ca, 0, sizeof(ca));
    for (int l = 0; l < nlayers; ++l) {
        const dcf_chain_layer &L = layers[l];
        DCF_REQUIRE(L.x && L.w && L.y, "dcf_conv3x3_chain: layer %d: null pointer", l);
        DCF_REQUIRE(l == 0 || L.x == layers[l - 1].y, "dcf_conv3x3_chain: layer %d must read layer %d's output", l, l - 1);
        for (int m = 0; m < l; ++m)
            DCF_REQUIRE(L.y != layers[m].y && L.y != layers[m].x, "dcf_conv3x3_chain: layer %d writes a tensor layer %d still uses", l, m);
        for (int m = l; m < nlayers; ++m)
            DCF_REQUIRE((!L.res || L.res != layers[m].y) && (!L.mask || L.mask != layers[m].y),
                        "dcf_conv3x3_chain: layer %d's residual / mask is the output of layer %d, which runs later", l, m);
        ca.L[l].x = (const char *)L.x; ca.L[l].w = (const char *)L.w; ca.L[l].shift = L.shift; ca.L[l].res = (const char *)L.res;
        ca.L[l].mask = (const char *)L.mask; ca.L[l].y = (char *)L.y; ca.L[l].relu = L.relu;
    }
    const int64_t Q = (int64_t)B * H * (W + 2);
    RsArgs &a = ca.a;
    a.B = B; a.H = H; a.W = W; a.Ck = C; a.Cn = C; a.flip = flip;
    a.npt = p.npt; a.Q = (int)Q;
    a.mtiles = (int)((Q + 32 * p.npt - 1) / (32 * p.npt));
    a.xbytes = a.ybytes = (unsigned)((int64_t)B * H * W * C * 2);
    a.wbytes = (unsigned)((int64_t)C * 9 * C * 2);
    ca.nlayers = nlayers;
    static DcfOpt sp_o("CHAIN_SPINS"), aq_o("CHAIN_ACQUIRE");
    const char *sp = sp_o.str(), *aq = aq_o.str();
    ca.spin_limit = sp ? atoi(sp) : (1 << 20);            // ~1 us per poll: about a second before a workgroup gives up
    ca.acquire = aq ? atoi(aq) : 0;
    ca.ws = (int *)ws;
    const dim3 grid((unsigned)nwg);
    char name[96];
    // (profile name: the single-layer launches' "conv_fwd_bf16<rs2,4>" plus the chain length)
    snprintf(name, sizeof(name), "%s_%s<rs%d,%d,x%d>", flip ? "conv_dgrad" : "conv_fwd", dtype == DCF_F16 ? "f16" : "bf16", p.kind, p.npt, nlayers);
    const double flops = 2.0 * B * H * W * (double)C * C * 9.0 * nlayers;
    double bytes = 0;
    for (int l = 0; l < nlayers; ++l)
        bytes += (double)a.wbytes + (double)a.xbytes * (2 + (layers[l].res ? 1 : 0) + (layers[l].mask ? 1 : 0));
    hipStream_t s = S(stream);
#define DCF_RSC(T_)                                                                                                              \
    do {                                                                                                                         \
        if (p.kind == 0) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 5, 4, 2, 2, 1, false, true>), grid, dim3(512), 0, s, ca)); \
        else if (p.kind == 1) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 3, 2, 4, 2, 1, false, true>), grid, dim3(512), 0, s, ca)); \
        else DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 1, 2, 4, 6, 2, true, true>), grid, dim3(512), 0, s, ca)); \
    } while (0)
#ifdef RS_CHAIN_DEV      // (development builds: one instantiation, a tenth of the compile time)
    DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<bf16_t, 1, 1, 2, 4, 6, 2, true, true>), grid, dim3(512), 0, s, ca));
#else
    if (dtype == DCF_F16) DCF_RSC(f16_t); else DCF_RSC(bf16_t);
#endif
#undef DCF_RSC
    return DCF_OK;
}

#ifdef RS_STAMP
// (variant builds only; tools/chain_stamps.py)
extern "C" int dcf_rs_stamps_read(long long *dst, int *dims)
{
    dims[0] = RS_STAMP_WGS; dims[1] = RS_STAMP_ITEMS; dims[2] = 10;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rs_stamps), sizeof(g_rs_stamps)) == hipSuccess ? 0 : -1;
}
extern "C" int dcf_rs_stamps_clear(void)
{
    static long long zero[RS_STAMP_WGS][RS_STAMP_ITEMS][10];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rs_stamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
