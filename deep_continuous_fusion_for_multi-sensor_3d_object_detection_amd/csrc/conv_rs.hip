// conv_rs.hip -- single-layer launches of the row-sharing 3x3 / stride-1 kernel (conv_rs_kernel.h has the kernel and its
// description; conv_chain.hip launches the same kernel over a chain of layers).
#include "conv_rs_kernel.h"

// Called by dcf_conv2d_fwd / dcf_conv2d_dgrad (conv.hip).  Returns DCF_EUNSUPPORTED when the shape is not this kernel's.
int dcf_conv3x3_rs_launch(int dtype, const void *x, const void *w, const float *shift, const void *res, const void *mask, void *y,
                          int B, int H, int W, int Ck, int Cn, int relu, int flip, const char *name_base, double flops, hipStream_t s)
{
    if (dtype == DCF_F32 || Ck % 64 || Cn % 64) return DCF_EUNSUPPORTED;
    const int64_t Q = (int64_t)B * H * (W + 2);
    if (Q >= (1ll << 30) || (int64_t)B * H * W * Ck * 2 >= (1ll << 31) || (int64_t)B * H * W * Cn * 2 >= 0xFFFFFF00ll) return DCF_EUNSUPPORTED;
    const RsPlan p = rs_plan(Q, Cn);
    if (p.kind < 0) return DCF_EUNSUPPORTED;
    RsArgs a;
    a.x = (const char *)x; a.w = (const char *)w; a.shift = shift; a.res = (const char *)res; a.mask = (const char *)mask; a.y = (char *)y;
    a.B = B; a.H = H; a.W = W; a.Ck = Ck; a.Cn = Cn; a.relu = relu; a.flip = flip;
    a.npt = p.npt; a.Q = (int)Q;
    a.dbg = dcf_ablate_opt("RS_DBG");
    a.mtiles = (int)((Q + 32 * p.npt - 1) / (32 * p.npt));
    a.xbytes = (unsigned)((int64_t)B * H * W * Ck * 2);
    a.wbytes = (unsigned)((int64_t)Cn * 9 * Ck * 2);
    a.ybytes = (unsigned)((int64_t)B * H * W * Cn * 2);
    const int BN = RS_KINDS[p.kind].BN;
    // persistent workgroups: at most one per CU, each walking its XCD's share of the tile list
    static DcfOpt pe_o("RS_PERSIST"); const char *pe = pe_o.str();
    int64_t nwg = (((int64_t)a.mtiles * (Cn / BN) + 7) / 8) * 8;
    if (!(pe && atoi(pe) == 0)) nwg = std::min<int64_t>(nwg, 256 * RS_KINDS[p.kind].per_cu);
    const dim3 grid((unsigned)nwg);
    char name[96];
    static DcfOpt pfn_o("RS_PF"); const char *pfn = pfn_o.str();
    static DcfOpt pfn2_o("RS_PF2"), pfl_o("RS_L16"), pfs_o("RS_S3");
    const bool small_pf = p.kind == 2 && !(pfn && atoi(pfn) == 0) && !(pfn2_o.str() && atoi(pfn2_o.str()) == 0) && !(pfl_o.str() && atoi(pfl_o.str()) == 0) &&
                          !(pfs_o.str() && atoi(pfs_o.str()) == 0);
    const bool one_round_n = (Q + 32 * p.npt - 1) / (32 * p.npt) * (Cn / RS_KINDS[p.kind].BN) <= 256 * RS_KINDS[p.kind].per_cu;
    const bool pf_name = small_pf || (p.kind != 2 && ((pfn && atoi(pfn) == 2) || (!(pfn && atoi(pfn) == 0) && (Ck >= 128 || one_round_n))));       // (= the loop chosen below: the profile name says which one ran)
    snprintf(name, sizeof(name), pf_name ? "%s<rs%d,%d,pf>" : "%s<rs%d,%d>", name_base, p.kind, p.npt);
    const double bytes = (double)a.xbytes + (double)a.wbytes + (double)B * H * W * Cn * 2.0 * (1 + (res ? 1 : 0) + (mask ? 1 : 0));
    static DcfOpt s3e_o("RS_S3"); const char *s3e = s3e_o.str();
    const bool s3 = !(s3e && atoi(s3e) == 0);
    static DcfOpt l16_o("RS_L16"); const char *l16e = l16_o.str();
    // consumer + loader waves (conv_rs_kernel.h).  RS_L16: the small-M kind with eight loader waves (default on).  (The other two
    // kinds were built with four loader waves -- twelve waves leave a wave the 170 registers their 48-80 accumulator registers
    // need; with sixteen they spill -- and measured no faster: DESIGN.md section 9; the instantiations are not kept.)
    const bool l16 = !(l16e && atoi(l16e) == 0);
    // kind 1 with at most eight position tiles per workgroup: the same tile shape instantiated for <= 256 positions (TMMAX = 2), whose
    // smaller pixel slot leaves LDS for a THIRD one -- the pixel tile of a stage then goes out two stages ahead (DX = 2) instead of
    // during the previous stage's first taps, 1-3 taps (~0.4-1.2 us) before its first use: those pixels come from HBM / the fabric,
    // and the loop waited for them at every stage.  Option RS_DX2=0: the DX = 1 instantiation for every launch of the kind.
    static DcfOpt dx2_o("RS_DX2"); const char *dx2e = dx2_o.str();
    const bool dx2 = p.kind == 1 && p.npt <= 8 && !(dx2e && atoi(dx2e) == 0);
    // RS_PF (round 6, default on): the rotated, fragment-prefetching tap loop for the two per-tap-synchronised kinds (conv_rs_kernel.h)
    static DcfOpt pf_o("RS_PF"); const char *pfe = pf_o.str();
    // (Ck = 64: three stages of HBM-bound taps -- the earlier synchronisation point of the rotated loop leaves the DMA a quarter tap less
    // to land: 64 -> 64 @352x400 33.3 -> 37.7 us; those layers keep the plain loop.  RS_PF=2 forces the rotated loop everywhere.)
    // A 64-channel launch of a single round of workgroups (the camera trunk's layer 1: 94x311) is latency-shaped like the wide ones
    // and takes the rotated loop too: 11.5 -> 10.5 us.
    const bool one_round = (int64_t)a.mtiles * (Cn / RS_KINDS[p.kind].BN) <= 256 * RS_KINDS[p.kind].per_cu;
    const bool pf = (pfe && atoi(pfe) == 2) || (!(pfe && atoi(pfe) == 0) && (Ck >= 128 || one_round));
    // the small-M kind's consumers on the rotated loop (stage-granular): RS_PF2 (default on with RS_PF)
    static DcfOpt pf2_o("RS_PF2"); const char *pf2e = pf2_o.str();
    const bool pf2 = p.kind == 2 && !(pfe && atoi(pfe) == 0) && !(pf2e && atoi(pf2e) == 0);
#define DCF_RS(T_)                                                                                                               \
    do {                                                                                                                         \
        if (p.kind == 0 && pf) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 5, 4, 2, 2, 1, false, false, 0, true>), grid, dim3(512), 0, s, a)); \
        else if (dx2 && pf) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 2, 2, 4, 2, 2, false, false, 0, true>), grid, dim3(512), 0, s, a)); \
        else if (p.kind == 1 && pf) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 3, 2, 4, 2, 1, false, false, 0, true>), grid, dim3(512), 0, s, a)); \
        else if (p.kind == 0) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 5, 4, 2, 2, 1>), grid, dim3(512), 0, s, a)); \
        else if (dx2) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 2, 2, 4, 2, 2>), grid, dim3(512), 0, s, a)); \
        else if (p.kind == 1) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 3, 2, 4, 2, 1>), grid, dim3(512), 0, s, a)); \
        else if (!s3) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 1, 2, 4, 5, 2>), grid, dim3(512), 0, s, a)); \
        else if (l16 && pf2) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 1, 2, 4, 6, 2, true, false, 8, true>), grid, dim3(1024), 0, s, a)); \
        else if (l16) DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 1, 2, 4, 6, 2, true, false, 8>), grid, dim3(1024), 0, s, a)); \
        else DCF_LAUNCH_WB(name, flops, bytes, s, hipLaunchKernelGGL((k_conv3x3_rs<T_, 1, 1, 2, 4, 6, 2, true>), grid, dim3(512), 0, s, a)); \
    } while (0)
#ifdef RS_BF16_ONLY            /* tools/rw_variants.sh: half the compile time */
    if (dtype == DCF_F16) return DCF_EUNSUPPORTED;
    DCF_RS(bf16_t);
#else
    if (dtype == DCF_F16) DCF_RS(f16_t); else DCF_RS(bf16_t);
#endif
#undef DCF_RS
    return DCF_OK;
}


#ifdef RS_WSTAMP
// variant builds only (tools/rs_wstamps.py): the per-wave barrier stamps of the last single-layer launch
extern "C" int dcf_rs_wstamps_read(void *dst, int *dims)
{
    dims[0] = RS_WS_WGS; dims[1] = 8; dims[2] = 3; dims[3] = 64;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rs_wstamps), sizeof(g_rs_wstamps)) == hipSuccess ? 0 : -1;
}
extern "C" int dcf_rs_wstamps_clear()
{
    static unsigned zero[RS_WS_WGS * 8 * 3 * 64];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rs_wstamps), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
