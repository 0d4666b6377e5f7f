/* conv_rw.h -- entry points of the EXPERIMENTAL register-weight 3x3 convolution (tools/variants/conv_rw.hip).
 * Not part of libdcf_hip.so and not declared in include/dcf_hip.h: `bash tools/rw_variants.sh rw=""` links the kernel beside
 * the product's objects into <pkg>/libdcf_hip_vrw.so for tools/rw_bench.py, rw_time.py and rw_ablate.py (DCF_HIP_LIB selects
 * it; tools/variants/rw_api.py binds the four symbols).  Round 4 measured it level with k_conv3x3_rs (profiles/r04b_*), so
 * the engine never used it; round 5 took it out of the shipped library.
 *
 * Weights in MFMA-fragment order: block (channel tile ct, tap, 64-channel chunk cc, k-step ks) = 1 KiB = 64 lanes x 16 bytes
 * at byte offset (((ct * 9 + tap) * (Cin / 64) + cc) * 4 + ks) * 1024; lane (r = lane % 32, h = lane / 32) holds output
 * channel 32 ct + r, input channels 64 cc + 16 ks + 8 h + {0..7} of that tap.  dcf_conv3x3_weight_frag converts
 * w [Cout][3][3][Cin] (what dcf_weight_prep writes); for the input gradient pass the fragment image of wt [Cin][3][3][Cout]. */
#pragma once
#include "../../include/dcf_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
int dcf_conv3x3_wf_supported(int dtype, int B, int H, int W, int Cin, int Cout);
int dcf_conv3x3_weight_frag(int dtype, const void *w, void *wf, int Cout, int Cin, dcf_stream_t stream);
int dcf_conv3x3_fwd_wf(int dtype, const void *x, const void *wf, const float *shift, const void *res, void *y,
                       int B, int H, int W, int Cin, int Cout, int relu, dcf_stream_t stream);
int dcf_conv3x3_dgrad_wf(int dtype, const void *gy, const void *wtf, const void *res, const void *mask, void *gx,
                         int B, int H, int W, int Cin, int Cout, dcf_stream_t stream);
#ifdef __cplusplus
}
#endif
