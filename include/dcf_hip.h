/*
 * dcf_hip.h -- C ABI of libdcf_hip.so, the MI355X (gfx950) implementation of the
 * continuous-fusion train-step hot path.
 *
 * The reference (Chanuk-Yang/Deep_Continuous_Fusion_for_Multi-Sensor_3D_Object_Detection)
 * is pure Python on torch: it has no FFI of its own, so the drop-in boundary is the
 * Python module surface (model.py / data_import_carla.py / loss.py / train.py) and THIS
 * header is what that surface binds underneath, through ctypes
 * (deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd/_hip.py).
 * Every entry point names the reference code it replaces (file:line under /root/reference).
 *
 * Conventions:
 *   - plain pointers and sizes; device pointers unless marked HOST; no torch types
 *   - returns 0 on success or a negative DCF_E* code; dcf_last_error() gives the text
 *   - never allocates, frees or synchronises; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*); workspaces are caller-provided
 *   - activations are NHWC ("pixel rows of channels"); dtype DCF_F32, DCF_BF16 or DCF_F16 (IEEE half),
 *     accumulation is always fp32
 *   - counts that are produced on the device (n_valid ...) stay on the device
 */
#ifndef DCF_HIP_H
#define DCF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *dcf_stream_t; /* hipStream_t */

enum { DCF_OK = 0, DCF_EINVAL = -1, DCF_ELAUNCH = -2, DCF_EUNSUPPORTED = -3 };
enum { DCF_F32 = 0, DCF_BF16 = 1, DCF_F16 = 2 };   /* compute / storage type of activations and weight images */
/* COMPAT: the reference's last-writer-wins scatter (data_import_carla.py:236-258), three launches; COMPAT_ROUNDS: the same
 * result by the literal nine rounds (claim pass c, resolve pass c-1), kept as a cross-check; ACCUM: atomic accumulation. */
enum { DCF_VOXEL_COMPAT = 0, DCF_VOXEL_ACCUM = 1, DCF_VOXEL_COMPAT_ROUNDS = 2,
       DCF_VOXEL_OCCUPANCY = 3 /* interpolate=False: voxel of the trunc'd ids := 1 (data_import_carla.py:231-234) */ };
enum { DCF_PROJ_COMPAT = 0, DCF_PROJ_CORRECT = 1 };

const char *dcf_last_error(void);
int dcf_version(void);
/* Tuning options (which kernel / tile shape a launch takes; never the results): each is seeded once per process from the
 * environment variable DCF_<NAME> and changed afterwards only here (value NULL = unset).  Tests use it to compare two
 * kernels on the same input (e.g. "KNN_KERNEL" = "tile" | "wave").  HOST strings. */
int dcf_set_option(const char *name, const char *value);
/* Optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg). */
int dcf_prof_enable(int on);
int dcf_prof_reset(void);
/* Records n empty brackets named "__empty_bracket__": the per-launch cost of the event pair itself. */
int dcf_prof_calibrate(dcf_stream_t stream, int n);
/* Fills up to `cap` records (one per kernel name = template instantiation); `work` receives the summed
 * ALGORITHMIC flops the launches declared (0 for kernels priced in bytes). Returns the number of
 * records. HOST pointers. */
int dcf_prof_read(char *names /*[cap][64]*/, double *total_ms, int64_t *calls, double *work, int cap);
/* Same, plus the algorithmic BYTES the launches declared (0 for launches that declare flops only). */
int dcf_prof_read2(char *names, double *total_ms, int64_t *calls, double *work, double *bytes, int cap);

/* ------------------------------------------------------------------ geometry
 * Replaces CarlaDataset.Voxelization_Projection / Projection
 * (data_import_carla.py:196-267, constants :31-43).                          */

/* Range filter + order-preserving compaction (data_import_carla.py:215-229).
 * lim HOST float[6] = {xlo,xhi,ylo,yhi,zlo,zhi}, strict inequalities.
 * out_pts [n][3], out_src [n] (may be NULL), count_dev int[1].
 * ws: dcf_compact_workspace_bytes(n). */
size_t dcf_compact_workspace_bytes(int n);
int dcf_range_filter(const float *pts, int n, const float *lim, float *out_pts, int32_t *out_src,
                     int32_t *count_dev, void *ws, dcf_stream_t stream);

/* Trilinear voxeliser (data_import_carla.py:236-258). Raw points in, range filter fused.
 * aff HOST float[6] = {sx,ox,sy,oy,sz,oz}.  grid [Cz][L][W] fp32 is fully written.
 * mode COMPAT: bit-exact "last writer wins per corner pass" (SURVEY.md F3);
 *      ACCUM : atomic trilinear splat.
 * owner_ws: int32[2][Cz*L*W], must be ZERO on entry and is returned ZERO (COMPAT only). */
size_t dcf_voxelize_workspace_bytes(int Cz, int L, int W);
int dcf_voxelize(const float *pts, int n, const float *lim, const float *aff, int Cz, int L, int W,
                 int mode, float *grid, void *owner_ws, dcf_stream_t stream);
/* Compat-mode voxeliser for the B frames of a batch (claim / gather / release, one launch each for all frames).  pts / n: HOST arrays of B device pointers / point counts; grids [B][Cz][L][W];
 * owner_ws: B * dcf_voxelize_workspace_bytes(Cz,L,W), zero on entry, returned zero.  Same results as B dcf_voxelize calls. */
#define DCF_MAX_VOXEL_BATCH 8
int dcf_voxelize_batch(const float *const *pts, const int *n, int B, const float *lim, const float *aff, int Cz, int L, int W,
                       float *grids, void *owner_ws, dcf_stream_t stream);
/* Same grids written as the engine's input image: x_nhwc [B][L][W][Cz] in `dtype` (what dcf_nchw_to_nhwc makes of the fp32
 * grids, bit for bit: a voxel's value is produced in one piece and rounded once).  Saves the fp32 grid and its transpose. */
int dcf_voxelize_batch_nhwc(int dtype, const float *const *pts, const int *n, int B, const float *lim, const float *aff, int Cz, int L,
                            int W, void *x_nhwc, void *owner_ws, dcf_stream_t stream);

/* Pinhole projection + in-image filter + compaction (data_import_carla.py:196-210,:261-266).
 * Raw points in; keeps points passing the range filter AND the image test, in order.
 * crt HOST float[12] = CRT_tensor [4][3].  uv_out [n][2], xyz_out [n][3], src_out [n] or NULL.
 * Rows >= *count_dev are left untouched (caller pre-zeroes for the zero padding of :263-266). */
int dcf_project_filter(const float *pts, int n, const float *lim, const float *crt, float ulim, float vlim,
                       int mode, float *uv_out, float *xyz_out, int32_t *src_out, int32_t *count_dev,
                       void *ws, dcf_stream_t stream);
/* The same for the B (<= 8) frames of a batch in one launch per phase (count / scan / scatter), every frame with its own points,
 * count and matrix: pts / n HOST arrays of B device pointers / point counts, crt HOST float[B][12]; uv_out [B][rows][2], xyz_out
 * [B][rows][3] (rows >= every n[b]), count_dev [B]; ws: B * dcf_compact_workspace_bytes(max n).  Bit-identical to B calls. */
int dcf_project_filter_batch(const float *const *pts, const int *n, int B, const float *lim, const float *crt, float ulim, float vlim,
                             int mode, float *uv_out, float *xyz_out, int rows, int32_t *count_dev, void *ws, dcf_stream_t stream);

/* BEV K-nearest-neighbour (reference: model.py:199-203 TODO; spec SURVEY.md App. D).
 * xyz [n_max][3], first *count_dev rows valid.  idx_out int32 [K][h][w], -1 padding.
 * rmax2 < 0 = unbounded.  K <= 8.  ws: dcf_knn_workspace_bytes(n_max,h,w). */
size_t dcf_knn_workspace_bytes(int n_max, int h, int w);
int dcf_knn_bev(const float *xyz, const int32_t *count_dev, int n_max, int K, int h, int w, int stride,
                float xs, float xo, float ys, float yo, float rmax2, int32_t *idx_out, void *ws,
                dcf_stream_t stream);
/* The B frames of a batch in one launch per phase: xyz [B][n_max][3], count_dev [B], idx_out [B][K][h][w]; ws = B workspaces of
 * dcf_knn_workspace_bytes(n_max, h, w), ws_stride_bytes apart (a multiple of 16).  Same indices as B calls of dcf_knn_bev. */
int dcf_knn_bev_batch(const float *xyz, const int32_t *count_dev, int B, int n_max, int K, int h, int w, int stride,
                      float xs, float xo, float ys, float yo, float rmax2, int32_t *idx_out, void *ws, size_t ws_stride_bytes,
                      dcf_stream_t stream);

/* A coarser site of the same batch (stride a multiple of fine_stride): its own cell sort as dcf_knn_bev_batch (ws), and a search
 * that serves the pixels of dense regions from the cells a FINER site's dcf_knn_bev_batch call has already built (ws_fine = that
 * call's workspace, untouched since; same xyz / count_dev / n_max / B) -- a few dozen candidates per pixel instead of the
 * hundreds to thousands a coarse cell window holds near the sensor.  Same indices as dcf_knn_bev_batch. */
int dcf_knn_bev_batch_shared(const float *xyz, const int32_t *count_dev, int B, int n_max, int K, int h, int w, int stride,
                             int fine_h, int fine_w, int fine_stride, float xs, float xo, float ys, float yo, float rmax2,
                             int32_t *idx_out, void *ws, size_t ws_stride_bytes, const void *ws_fine, size_t ws_fine_stride_bytes,
                             dcf_stream_t stream);

/* All fusion sites of a batch in one call: the cell sort of every site in one launch per phase, then each site's search as
 * dcf_knn_bev_batch (fine = -1) or dcf_knn_bev_batch_shared (fine = index of an earlier, finer site of the call) runs it: the
 * same maps bit for bit, 6 sort launches instead of 6 per site.  The reference computes these indices once per frame in its
 * loader (data_import_carla.py: the commented KNN block).  sites is a HOST array of 1..4 entries; every ws is
 * 16-byte aligned with frames ws_stride_bytes apart (>= dcf_knn_workspace_bytes(n_max, h, w), multiple of 16). */
typedef struct { int32_t h, w, stride, fine; int32_t *idx_out; void *ws; size_t ws_stride_bytes; } dcf_knn_site;
int dcf_knn_bev_sites(const float *xyz, const int32_t *count_dev, int B, int n_max, int K, const dcf_knn_site *sites, int nsites,
                      float xs, float xo, float ys, float yo, float rmax2, dcf_stream_t stream);

/* Inverse of the KNN maps of a step (sites x frames) for the fusion backward: the (pixel, point) pairs of every
 * idx [K][h][w] counting-sorted by (map, point).  start int32 [nmaps*(n_max+1)]: pairs of point q of map g are
 * [start[g*(n_max+1)+q], start[g*(n_max+1)+q+1]); ent_pix / ent_pt int32 [sum K*h*w] (pixel packed (i<<16)|j).
 * maps is a HOST array.  ws: dcf_fusion_invert_workspace_bytes(n_max, nmaps).  Pair order inside a point is not
 * deterministic (atomic cursor). */
#define DCF_MAX_KNN_MAPS 32
typedef struct { const int32_t *idx; int32_t h, w; } dcf_knn_map;
size_t dcf_fusion_invert_workspace_bytes(int n_max, int nmaps);
int dcf_fusion_invert(const dcf_knn_map *maps, int nmaps, int K, int n_max, int32_t *start, int32_t *ent_pix, int32_t *ent_pt,
                      void *ws, dcf_stream_t stream);

/* ------------------------------------------------------------ layout / input
 * NCHW fp32 -> NHWC dtype (the model keeps the reference's NCHW voxel input, model.py:194). */
int dcf_nchw_to_nhwc(int dtype, const float *x, void *y, int B, int C, int H, int W, dcf_stream_t stream);
/* NHWC dtype -> NCHW fp32: stage outputs handed back in the reference's layout (model.py:76-79). */
int dcf_nhwc_to_nchw(int dtype, const void *x, float *y, int B, int C, int H, int W, dcf_stream_t stream);
/* uint8 NCHW image -> x/255 as NHWC with C padded to 4 and a zero halo of 3 pixels (stem input). */
int dcf_image_to_nhwc4(int dtype, const uint8_t *img, void *y, int B, int H, int W, dcf_stream_t stream);

/* -------------------------------------------------------------- convolution
 * Replaces nn.Conv2d (+ folded eval BatchNorm + residual + ReLU) of model.py:15-41,147-157.
 * Implicit GEMM on MFMA; x [B,H,W,Cin], w [Cout][kh][kw][Cin] (dtype), y [B,Ho,Wo,Cout].
 * y = act(conv(x,w) + shift[c] + res);  shift fp32 [Cout] or NULL; res (dtype, like y) or NULL.
 * Cin*esize must be a multiple of 64 bytes; Cout a multiple of 32. */
int dcf_conv2d_fwd(int dtype, const void *x, const void *w, const float *shift, const void *res, void *y,
                   int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                   int relu, dcf_stream_t stream);
/* 1x1 layers: the shift scaled per output pixel, y = act(conv(x,w) + rowscale[m]*shift[c] + res); rowscale fp32 [B*Ho*Wo].
 * The bias of the fusion site's second Linear layer under the neighbour sum (reference model.py:216-219:
 * sum_k (W2 h_k + b2) = W2 sum_k h_k + cnt*b2), in the GEMM's epilogue. */
int dcf_conv2d_fwd_rowscale(int dtype, const void *x, const void *w, const float *shift, const float *rowscale, const void *res,
                            void *y, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                            int relu, dcf_stream_t stream);
/* Input gradient: gx [B,H,W,Cin] = conv_transpose(gy [B,Ho,Wo,Cout], wt) (+ res).
 * wt [Cin][kh][kw][Cout] (dtype) as produced by dcf_weight_prep.
 * Optional fused ReLU backward of the layer that produced x: mask (dtype, like gx) zeroes gx where
 * mask <= 0. */
int dcf_conv2d_dgrad(int dtype, const void *gy, const void *wt, const void *res, const void *mask, void *gx,
                     int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                     dcf_stream_t stream);
/* The same for a stride-2 layer with a kernel wider than 1x1, plus a residual that lives on the (2i, 2j)
 * sub-grid of gx: resq [B,ceil(H/2),ceil(W/2),Cin] (dtype).  gx = (dgrad + res + scatter(resq)) * (mask > 0).
 * resq is the input gradient of the block's 1x1 / stride-2 shortcut (reference model.py:27-30, 38-40),
 * computed on its own grid by dcf_conv2d_dgrad(..., H = ceil(H/2), W = ceil(W/2), stride = 1). */
int dcf_conv2d_dgrad_halfres(int dtype, const void *gy, const void *wt, const void *res, const void *resq,
                             const void *mask, void *gx, int B, int H, int W, int Cin, int Ho, int Wo, int Cout,
                             int kh, int kw, int stride, int pad, dcf_stream_t stream);
/* ---- A CHAIN of 3x3 / stride-1 / pad-1 layers with C channels in and out, each reading the one before it, in ONE launch
 * (csrc/conv_rs.hip, chain mode): the bodies of a residual stage -- /root/reference/model.py:32-41, conv1 -> bn1 -> relu ->
 * conv2 -> bn2 -> += shortcut -> relu, block after block (model.py:48-60) -- forward, or (flip = 1, weights = the
 * [Cin][tap][Cout] images) the same list backwards for the input gradients.  Layer l computes exactly what
 *   dcf_conv2d_fwd(dtype, x, w, shift, res, y, B, H, W, C, H, W, C, 3, 3, 1, 1, relu)            (flip = 0)
 *   dcf_conv2d_dgrad(dtype, x, w, res, mask, y, B, H, W, C, H, W, C, 3, 3, 1, 1)                 (flip = 1; shift / relu unused)
 * computes with its row-sharing kernel -- bit-identical results -- but a workgroup moves from its tile of layer l to its tile of
 * layer l + 1 as soon as the neighbouring tiles of layer l are complete (per-tile arrival counters, write-through stores, sc1
 * loads: no kernel boundary, no grid-wide wait).  layers is a HOST array; layers[l].x must be layers[l - 1].y; res may be
 * any tensor written before the launch or the y of a layer at least two back; outputs must be distinct tensors.
 * ws: dcf_conv3x3_chain_workspace_bytes(...) bytes, ZEROED ONCE by the caller when allocated (the launch leaves it zero);
 * one workspace per stream of launches.  dcf_conv3x3_chain_supported: 16-bit dtypes, C % 64 == 0, a layer's tiles fit one
 * round of workgroups (option CONV_CHAIN=0 answers no).  A workgroup whose bounded wait expires writes a record to
 * ((int32_t *)ws)[1] (0 = none; 0x40000000 | layer << 16 | position tile) and stops waiting: results are then wrong, the GPU
 * is not hung; callers that want to know copy that word back (tests do after every launch). */
#define DCF_CHAIN_MAX_LAYERS 24
typedef struct {
    const void *x, *w;
    const float *shift;        /* fp32 [C] or NULL */
    const void *res, *mask;    /* dtype [B,H,W,C] or NULL */
    void *y;
    int32_t relu, pad_;
} dcf_chain_layer;
int dcf_conv3x3_chain_supported(int dtype, int B, int H, int W, int C, int nlayers);
size_t dcf_conv3x3_chain_workspace_bytes(int dtype, int B, int H, int W, int C, int nlayers);
int dcf_conv3x3_chain(int dtype, const dcf_chain_layer *layers, int nlayers, int B, int H, int W, int C, int flip,
                      void *ws, dcf_stream_t stream);
/* Weight gradient, split over pixel ranges: slabs fp32 [nsplit][Cout][kh][kw][Cin] (plain stores,
 * reduced in fixed order by dcf_wgrad_finalize => bitwise reproducible).
 * gsum (optional) fp32 [4*nsplit][Cout]: per-wave sums over pixels of gy (dL/dbeta of a folded BN),
 * accumulated by the same kernel from the gy fragments it already holds.
 * nsplit = dcf_conv2d_wgrad_splits(...). */
int dcf_conv2d_wgrad_splits(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride);
int dcf_conv2d_wgrad(int dtype, const void *x, const void *gy, float *slabs, float *gsum, int nsplit,
                     int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                     dcf_stream_t stream);
/* The weight gradients of a backward pass are independent of each other: dcf_conv2d_wgrad_group takes all of them (a
 * HOST array) and issues them kernel class by kernel class, up to 32 layers per launch -- the next layer's workgroups
 * start as the previous one's finish (no drain / launch bubble, no idle tail).  Layers of the LDS-DMA row-sharing kernel
 * (dcf_conv2d_wgrad_groupable() == 1: 3x3 / stride 1 / pad 1, 16-bit dtype, >= 64 channels on both sides) and of the
 * generic kernel are grouped; the remaining ones (fp32, 32-channel row-sharing layers) are launched one by one.
 * x / gy must stay alive and unmodified until the call.  Same slabs / gsum / nsplit as dcf_conv2d_wgrad. */
typedef struct {
    int32_t dtype, nsplit;
    const void *x, *gy;
    float *slabs, *gsum;       /* gsum may be NULL */
    int32_t B, H, W, Cin, Cout, kh, kw, stride, pad, pad_;
} dcf_wgrad_item;
int dcf_conv2d_wgrad_groupable(int dtype, int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
int dcf_conv2d_wgrad_group(const dcf_wgrad_item *items, int n, dcf_stream_t stream);
/* The 7x7/2 RGB stem of the image stream on the NHWC4+halo image. */
int dcf_stem7x7_fwd(int dtype, const void *img4, const void *w, const float *shift, void *y,
                    int B, int H, int W, int Ho, int Wo, int Cout, int relu, dcf_stream_t stream);
int dcf_stem7x7_wgrad(int dtype, const void *img4, const void *gy, float *slabs, float *gsum, int nsplit,
                      int B, int H, int W, int Ho, int Wo, int Cout, dcf_stream_t stream);

/* Per-step parameter preparation (table driven, one launch for the whole net):
 * for conv i: scale = gamma*rsqrt(var+eps) (or 1), shift = beta - mean*scale (or 0),
 * w_fwd = cast(scale[co]*W), w_dgrad = transpose of the same.  Descriptor table lives on
 * the device (struct dcf_conv_param, below).  cin and cout_pad must be multiples of 32. */
typedef struct dcf_conv_param {
    int64_t w_off;      /* element offset of W [Cout][taps][Cin] fp32 in the parameter arena */
    int64_t gamma_off;  /* BN gamma/beta offsets in the parameter arena, -1 = no BN            */
    int64_t beta_off;
    int64_t mean_off;   /* running mean/var offsets in the buffer arena                        */
    int64_t var_off;
    int64_t wfwd_off;   /* byte offsets into the compute-dtype weight arena                    */
    int64_t wdgrad_off; /* -1 = not needed                                                      */
    int64_t shift_off;  /* element offset into the fp32 scale/shift arena: [scale Cout][shift Cout] */
    int64_t slab_off;   /* element offset of this conv's wgrad slabs in the slab arena         */
    int64_t gsum_off;   /* element offset of this conv's [4*nsplit][cout_pad] sums of g in the gsum arena */
    int32_t cout, cin, taps, cout_pad;
    int32_t nsplit, flags, pad0, pad1;
} dcf_conv_param;
int dcf_weight_prep(int dtype, const dcf_conv_param *table, int nconv, const float *params, const float *buffers,
                    void *warena, float *ssarena, float eps, dcf_stream_t stream);
/* Reduce wgrad slabs in split order and apply the folded-BN chain rule:
 * dW = scale*G, dgamma = (<W,G> - mean*dbeta)*invstd, dbeta = sum over the 4*nsplit rows of gsum[.][co]. */
/* max_cout = the largest cout of the table (grid width). */
int dcf_wgrad_finalize(const dcf_conv_param *table, int nconv, int max_cout, const float *params, const float *buffers,
                       const float *ssarena, const float *slabs, const float *gsum, float *grads, float eps,
                       dcf_stream_t stream);
/* Same, with the layers' cout values on the HOST (cout_host[nconv], nconv <= 256): one workgroup per (conv, output channel)
 * exactly, instead of a max_cout x nconv grid whose surplus workgroups exit at once. */
int dcf_wgrad_finalize_rows(const dcf_conv_param *table, int nconv, const int32_t *cout_host, const float *params, const float *buffers,
                            const float *ssarena, const float *slabs, const float *gsum, float *grads, float eps,
                            dcf_stream_t stream);

/* ------------------------------------------------------------- fp8 forward convolutions (csrc/conv_fp8.hip)
 * BASELINE.json configs[4] / SURVEY.md 8(d) cfg5 ("fp8 MFMA convs, fp32 accumulate, bf16 epilogue"); the reference has
 * no counterpart (model.py runs fp32 nn.Conv2d, model.py:16-19).  Operands are OCP e4m3 on
 * v_mfma_scale_f32_32x32x64_f8f6f4; weights carry one scale per output channel, activations one power-of-two scale per
 * tensor derived on the device from the tensor's absolute maximum one step earlier (delayed scaling, no host sync):
 *   y = act((sum_k q(x*sx) q(w*sw[co])) / (sx*sw[co]) + shift[co] + res)
 * dcf_fp8_act_scale: the scale rule (largest power of two s with amax*s <= 224; 1 when amax is 0 / not finite). */
int dcf_fp8_act_scale(float amax, float *scale);
/* x8[i] = q(x[i] * dcf_fp8_act_scale(*amax_prev)) (amax_prev null: scale 1).  amax_cur (optional): 64 device floats
 * holding partial maxima of |x| -- each workgroup raises one of them, so the tensor's maximum is the max over the 64
 * (one hot address would serialise the atomics).  n multiple of 8. */
int dcf_cast_fp8(int dtype, const void *x, void *x8, const float *amax_prev, float *amax_cur, int64_t n, dcf_stream_t stream);
/* Per-conv fp8 weight images: w8 [cout_pad][taps][cin] = q(bn_scale*W * 448/amax_co) at byte offset w8_off of w8arena,
 * dequantisation factors amax_co/448 at element offset wscale_off of wsarena (w8_off < 0: conv not on the fp8 path).
 * Also rolls each conv's activation maxima: amax holds DCF_F8_AMAX_STRIDE floats per conv, [0..63] the partial maxima of
 * the current step and [64] the previous step's maximum (the scale source): prev <- max(cur[0..63]), cur <- 0. */
#define DCF_F8_AMAX_STRIDE 80
typedef struct dcf_f8_param {
    int64_t w8_off;
    int64_t wscale_off;
} dcf_f8_param;
int dcf_weight_prep_fp8(const dcf_conv_param *table, const dcf_f8_param *f8table, int nconv, int max_cout_pad, const float *params,
                        const float *buffers, void *w8arena, float *wsarena, float *amax, float eps, dcf_stream_t stream);
/* Forward convolution over fp8 images: x8 [B][H][W][Cin], w8 [Cout][kh][kw][Cin], wscale [Cout], xamax = the device
 * scalar dcf_cast_fp8 derived x8's scale from (null: 1); shift / res / relu as dcf_conv2d_fwd; y, res in out_dtype.
 * Optional second output for the convolution that consumes y: y8 = dcf_cast_fp8(y) with the scale of *y8amax, partial
 * maxima of |y| raised in y8cur[64] (either may be null).  Cin multiple of 64, Cout multiple of 32. */
int dcf_conv2d_fwd_fp8(int out_dtype, const void *x8, const void *w8, const float *wscale, const float *xamax, const float *shift,
                       const void *res, void *y, void *y8, const float *y8amax, float *y8cur, int B, int H, int W, int Cin,
                       int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int relu, dcf_stream_t stream);

/* ------------------------------------------------------------- elementwise
 * g = gy * (y > 0) in place on gy (ReLU backward, model.py:21,25) and per-channel sums
 * gsum[c] += sum_p g[p][c] (the BN-beta gradient).  gsum fp32 [C], pre-zeroed by caller. */
int dcf_relu_bwd_chansum(int dtype, void *gy, const void *y, float *gsum, int64_t npix, int C, int relu,
                         dcf_stream_t stream);
/* Bilinear resize NHWC (model.py:149,151 nn.UpsamplingBilinear2d => align_corners=1; the
 * image FPN uses align_corners=0).  y = resize(x) + (add ? add : 0). */
int dcf_resize_bilinear_fwd(int dtype, const void *x, const void *add, void *y, int B, int Hi, int Wi, int Ho, int Wo,
                            int C, int align_corners, dcf_stream_t stream);
/* gx = resize^T(gy): gather form (each input pixel sums its contributing output pixels). */
int dcf_resize_bilinear_bwd(int dtype, const void *gy, void *gx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                            int align_corners, dcf_stream_t stream);
/* 3x3/2 max-pool pad 1 (image stem), NHWC; bwd routes to the first arg-max in scan order. */
int dcf_maxpool3x3s2_fwd(int dtype, const void *x, void *y, int B, int H, int W, int Ho, int Wo, int C,
                         dcf_stream_t stream);
int dcf_maxpool3x3s2_bwd(int dtype, const void *x, const void *y, const void *gy, void *gx, int B, int H, int W,
                         int Ho, int Wo, int C, dcf_stream_t stream);
/* Same pooling with the arg-max recorded: idx uint32 [B][Ho][Wo][C/4], one byte per channel = window position dh*3+dw of
 * the first maximum in scan order; the backward is then a gather over (idx, gy) that never re-reads x. */
int dcf_maxpool3x3s2_fwd_idx(int dtype, const void *x, void *y, uint32_t *idx, int B, int H, int W, int Ho, int Wo, int C,
                             dcf_stream_t stream);
int dcf_maxpool3x3s2_bwd_idx(int dtype, const uint32_t *idx, const void *gy, void *gx, int B, int H, int W, int Ho, int Wo, int C,
                             dcf_stream_t stream);
/* Heads (model.py:168-172 softmax pairs, :116-137 box decode, :204 concat):
 * head [B,h,w,Cp] (first 18 channels = 4 class logits + 14 offsets) ->
 * pred [B,32,h,w] fp32 NCHW = cat(softmax2,softmax2, reg14, decode(reg14, anchors[14,h,w])). */
int dcf_head_fwd(int dtype, const void *head, int Cp, const float *anchors, float *pred, int B, int h, int w,
                 dcf_stream_t stream);
/* gpred [B,32,h,w] fp32 -> ghead [B,h,w,Cp] (dtype), pad channels zero. */
int dcf_head_bwd(int dtype, const void *head, int Cp, const float *anchors, const float *pred, const float *gpred,
                 void *ghead, int B, int h, int w, dcf_stream_t stream);

/* Train-mode BatchNorm2d (model.py:20,24,29 with the module in .train(): batch statistics, momentum 0.1,
 * eps 1e-5).  fwd: batch mean / invstd (fp32 [C], kept for backward), running stats updated in place
 * (unbiased variance) when non-NULL, y = act(gamma*(x-mean)*invstd + beta + res).
 * bwd: dgamma/dbeta written (not accumulated), dx = gamma*invstd*(g - dbeta/M - xhat*dgamma/M).
 * ws: dcf_bn_workspace_bytes(C). */
size_t dcf_bn_workspace_bytes(int C);
int dcf_bn_train_fwd(int dtype, const void *x, const float *gamma, const float *beta, const void *res, void *y,
                     float *mean, float *invstd, float *running_mean, float *running_var, int64_t npix, int C,
                     float eps, float momentum, int relu, void *ws, dcf_stream_t stream);
int dcf_bn_train_bwd(int dtype, const void *g, const void *x, const float *mean, const float *invstd, const float *gamma,
                     float *dgamma, float *dbeta, void *dx, int64_t npix, int C, void *ws, dcf_stream_t stream);

/* ------------------------------------------------------------------ fusion
 * SURVEY.md App. D (reference: model.py:199-203 TODO).  Per sample.
 * (1) per-point camera feature: fp [n][Cf] = bilinear(F [Hf][Wf][Cf], u/4-0.5, v/4-0.5), border clamp; all n_max rows of fp are
 *     written (zeros from *count_dev on: fp needs no clearing) */
int dcf_point_sample_fwd(int dtype, const void *fmap, int Hf, int Wf, int Cf, const float *uv, const int32_t *count_dev,
                         int n_max, void *fp, dcf_stream_t stream);
/*     backward: gF[tap] += w_tap * gfp (fp32 atomics into gfmap fp32 [Hf][Wf][Cf]) */
int dcf_point_sample_bwd(int dtype, const void *gfp, int Hf, int Wf, int Cf, const float *uv, const int32_t *count_dev,
                         int n_max, float *gfmap, dcf_stream_t stream);
/*     The B frames of a batch in one launch each (grid.y = frame): fmap / gfmap [B][Hf][Wf][Cf], uv frame b at uv + b * uv_fstride
 *     floats, count_dev [B], fp / gfp [B][n_max][Cf].  Same results as B single-frame calls. */
int dcf_point_sample_fwd_batch(int dtype, const void *fmap, int Hf, int Wf, int Cf, const float *uv, int64_t uv_fstride,
                               const int32_t *count_dev, int n_max, void *fp, int B, dcf_stream_t stream);
int dcf_point_sample_bwd_batch(int dtype, const void *gfp, int Hf, int Wf, int Cf, const float *uv, int64_t uv_fstride,
                               const int32_t *count_dev, int n_max, float *gfmap, int B, dcf_stream_t stream);
/* (2) per BEV pixel: hsum[p][c] = sum_k relu(P[idx_k][c] + W1d[c][0..2].(dx,dy,z) + b1[c]),
 *     cnt[p] = number of valid neighbours.  P [n][Cb] (dtype); w1d fp32 [Cb][3]; b1 fp32 [Cb]. */
int dcf_fusion_gather_fwd(int dtype, const void *P, const float *xyz, const int32_t *idx, int K, int h, int w,
                          int stride, float xs, float xo, float ys, float yo, const float *w1d, const float *b1,
                          int Cb, void *hsum, float *cnt, dcf_stream_t stream);
/*     batch form: P [B][p_rows][Cb], xyz frame b at xyz + b * xyz_fstride floats, idx [B][K][h][w], hsum [B][h][w][Cb], cnt [B][h*w] */
int dcf_fusion_gather_fwd_batch(int dtype, const void *P, int64_t p_rows, const float *xyz, int64_t xyz_fstride, const int32_t *idx,
                                int K, int h, int w, int stride, float xs, float xo, float ys, float yo, const float *w1d,
                                const float *b1, int Cb, void *hsum, float *cnt, int B, dcf_stream_t stream);
/*     backward: recompute the ReLU mask; gP[idx_k] += m*gh (fp32 atomics, gP fp32 [n][Cb]);
 *     gw1d[c][j] += sum m*gh*delta_j ; gb1[c] += sum m*gh. */
int dcf_fusion_gather_bwd(int dtype, const void *P, const float *xyz, const int32_t *idx, int K, int h, int w,
                          int stride, float xs, float xo, float ys, float yo, const float *w1d, const float *b1,
                          int Cb, const void *ghsum, float *gP, float *gw1d, float *gb1, dcf_stream_t stream);
/* y[p][c] += cnt[p]*b2[c]  (the fc2 bias of the K-sum), and its gradient gb2[c] += sum_p cnt[p]*gy[p][c]. */
int dcf_rowscale_bias_fwd(int dtype, void *y, const float *cnt, const float *b2, int64_t npix, int C, dcf_stream_t stream);
int dcf_rowscale_bias_bwd(int dtype, const void *gy, const float *cnt, float *gb2, int64_t npix, int C, dcf_stream_t stream);
/* (version 201) the same bias gradient and, in the same pass, the gradient that goes on into the stage's last block:
 *     gout[p][c] = y[p][c] > 0 ? gy[p][c] : 0 ;  gb2[c] += sum_p cnt[p]*gy[p][c]   (gout is its own tensor: the fusion branch still
 *     reads the unmasked gy).  Replaces dcf_rowscale_bias_bwd + the in-place dcf_relu_bwd_chansum at a fusion site. */
int dcf_relu_mask_rowscale_bwd(int dtype, const void *gy, const void *y, const float *cnt, void *gout, float *gb2, int64_t npix, int C,
                               dcf_stream_t stream);
/* dtype <-> fp32 casts of whole buffers (gradient hand-offs of the fusion path). */
int dcf_cast(int dtype_src, const void *src, int dtype_dst, void *dst, int64_t n, dcf_stream_t stream);

/* ------------------------------------------------------------------- optimiser
 * Fused Adam over the flat fp32 parameter arena (train.py:28,36: Adam(lr, betas=(beta1,0.999)),
 * eps 1e-8, no weight decay, bias-corrected like torch.optim.Adam). gscale multiplies the gradient
 * (1/world_size after the all-reduce). */
int dcf_adam_step(float *params, const float *grads, float *m, float *v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float gscale, dcf_stream_t stream);

/* dcf_fusion_gather_bwd driven by dcf_fusion_invert's pairs (Cb in {64,128,192,256}): same sums, no idx -> point -> row
 * dependency chain.  The map's pairs are [*e_begin, *e_end) = start[g*(n_max+1)], start[g*(n_max+1)+n_max];
 * max_entries = K*h*w sizes the grid.
 * workspace (optional, dcf_fusion_gather_bwd_workspace_bytes(Cb), ZERO before its first use, left zero = reusable): with it the
 * workgroups' sums of gw1d / gb1 go through 16 copies of the accumulators that the last-arriving workgroup folds (16 instead of
 * 256 same-address atomics per word); NULL: float atomics straight onto gw1d / gb1.  Calls that share a workspace must be
 * ordered on one stream. */
size_t dcf_fusion_gather_bwd_workspace_bytes(int Cb);
int dcf_fusion_gather_bwd_inv(int dtype, const void *P, const float *xyz, const int32_t *e_begin, const int32_t *e_end,
                              const int32_t *ent_pix, const int32_t *ent_pt, int max_entries, int h, int w, int stride, float xs,
                              float xo, float ys, float yo, const float *w1d, const float *b1, int Cb, const void *ghsum, float *gP,
                              float *gw1d, float *gb1, void *workspace, dcf_stream_t stream);
/* The B frames of a batch (consecutive maps of one dcf_fusion_invert call) in one launch: P / gP [B][p_rows][Cb], xyz frame b at
 * xyz + b * xyz_fstride floats, frame b's e_begin / e_end at + b * seg_fstride ints (= n_max + 1 for consecutive maps), ghsum
 * [B][h][w][Cb]; gw1d / gb1 accumulate over the frames.  B <= 64. */
int dcf_fusion_gather_bwd_inv_batch(int dtype, const void *P, int64_t p_rows, const float *xyz, int64_t xyz_fstride, const int32_t *e_begin,
                                    const int32_t *e_end, int64_t seg_fstride, const int32_t *ent_pix, const int32_t *ent_pt, int max_entries,
                                    int h, int w, int stride, float xs, float xo, float ys, float yo, const float *w1d, const float *b1, int Cb,
                                    const void *ghsum, float *gP, float *gw1d, float *gb1, void *workspace, int B, dcf_stream_t stream);

/* The same sums with gP [B][p_rows][Cb] in the COMPUTE dtype and ZERO on entry (rows of points no pixel chose stay zero): a
 * point whose pairs all sit in one 16..128-pair slice of the sorted list has one writer and its row is stored whole; a point
 * whose run crosses slices is summed in an fp32 row of direct_ws by the slices of its run, and the last of them (a ticket per
 * row) stores it.  direct_ws: dcf_fusion_gather_bwd_direct_workspace_bytes(max_entries, Cb, B) bytes, zero on entry, left zero.
 * No fp32 accumulator the size of gP to fill before the launch and to cast after it.  (Reference: the autograd of
 * model.py:210-219 -- index_select / cat / Linear / ReLU / sum -- for dL/d(point features).) */
size_t dcf_fusion_gather_bwd_direct_workspace_bytes(int max_entries, int Cb, int B);
int dcf_fusion_gather_bwd_direct_batch(int dtype, const void *P, int64_t p_rows, const float *xyz, int64_t xyz_fstride,
                                       const int32_t *e_begin, const int32_t *e_end, int64_t seg_fstride, const int32_t *ent_pix,
                                       const int32_t *ent_pt, int max_entries, int h, int w, int stride, float xs, float xo, float ys,
                                       float yo, const float *w1d, const float *b1, int Cb, const void *ghsum, void *gP, float *gw1d,
                                       float *gb1, void *workspace, void *direct_ws, int B, dcf_stream_t stream);

/* The same sums with ONE writer per point row (a wave owns a range of points and all their pairs): gP [n_rows][Cb] in the compute
 * dtype, every row written (zeros where no pixel chose the point) -- no zero-filled fp32 accumulator, no float atomics on gP, no
 * cast afterwards.  start = the map's slice of dcf_fusion_invert's start array (n_rows + 1 entries are read, n_rows <= n_max). */
int dcf_fusion_gather_bwd_pts(int dtype, const void *P, const float *xyz, const int32_t *start, int n_rows, const int32_t *ent_pix,
                              const int32_t *ent_pt, int max_entries, int h, int w, int stride, float xs, float xo, float ys, float yo,
                              const float *w1d, const float *b1, int Cb, const void *ghsum, void *gP, float *gw1d, float *gb1,
                              dcf_stream_t stream);

/* ------------------------------------------------------------- detection objective (loss.py:129-189)
 * Device half of LossTotal: 2-way cross-entropy at the sampled cells of both anchors + Smooth-L1 of the encoded box
 * offsets, and their gradients (fp32 atomics into ZEROED dense maps), in one launch; *loss (zeroed) receives the scalar.
 * cls [B][4][HW] / reg [B][14][HW] fp32 with the given batch strides (elements); anchors [2][7][HW].
 * ints (int64) = B x {off_int, npos, nneg, nrow, off_float, nbox}, then per sample: positive cells, negative cells,
 * regression cells, box index of each regression cell;  floats = per sample: weight of each regression cell, boxes [nbox][7].
 * The host-side target assignment that fills them is loss.py:74-127 (numpy RNG).  reduction: 0 last sample only
 * (reference behaviour), 1 sum, 2 mean over the batch. */
int dcf_loss_fwd_bwd(const float *cls, int64_t cls_bstride, const float *reg, int64_t reg_bstride, const float *anchors,
                     const int64_t *ints, const float *floats, int B, int HW, float reg_gain, int reduction,
                     float *loss, float *gcls, int64_t gcls_bstride, float *greg, int64_t greg_bstride, dcf_stream_t stream);

/* The same objective with the target assignment ON THE DEVICE (loss.py:74-127 without the host): per sample the
 * positive windows (span x span cells around each labelled box's centre cell, fp32 arithmetic as loss.py:85-86), a uniform subset
 * of pos_cap entries when there are more (loss.py:107-110), neg_count cells drawn with replacement and rejected against the
 * selected positives (loss.py:117-126), then the terms and gradients of dcf_loss_fwd_bwd -- one launch, one workgroup per sample.
 * boxes [B][max_box][box_stride] fp32 on the device (x, y, z, l, w, h, yaw first), nbox_dev int32 [B]; xs, xo, ys, yo = grid scale /
 * offset (data_import_carla.py:35-43), reduced_scale = anchor stride.  Randomness = dcf_loss_sample_rand(seed, sample, stream, index,
 * attempt), a stateless 64-bit mix (stream 1: subset keys, the pos_cap smallest (key, entry) win; stream 2: negative item `index`,
 * cell = (rand * H*W) >> 32).  Optional outputs for inspection: pos_out int32 [B][pos_cap] (-1 padded), neg_out [B][neg_count],
 * counts_out [B][2] = {selected positives, window entries}.  max_box <= 64, max_box * span^2 <= 1024, neg_count <= 512. */
uint32_t dcf_loss_sample_rand(uint64_t seed, int sample, int stream, int index, int attempt);
int dcf_loss_sample_fwd_bwd(const float *cls, int64_t cls_bstride, const float *reg, int64_t reg_bstride, const float *anchors,
                            const float *boxes, const int32_t *nbox_dev, int max_box, int box_stride, int B, int H, int W,
                            float xs, float xo, float ys, float yo, float reduced_scale, int span, int regress_type, int pos_cap,
                            int neg_count, uint64_t seed, float reg_gain, int reduction, float *loss, float *gcls,
                            int64_t gcls_bstride, float *greg, int64_t greg_bstride, int32_t *pos_out, int32_t *neg_out,
                            int32_t *counts_out, dcf_stream_t stream);

/* ------------------------------------------------- evaluation post-processing
 * What /root/reference/test.py:88-206 does on the host, box by box.
 * dcf_eval_score_filter: test.py:88-108.  pred [B][32][h][w] fp32 (the model output: scores in channels 2a+1, decoded boxes in
 *   18+7a..18+7a+6 for anchor a); boxes_out [B][cap][7] receives, per sample, anchor 0's boxes with score > threshold in raster
 *   order, then anchor 1's; count_out[b] = how many passed (may exceed cap: rows past cap are dropped).
 * dcf_eval_nms: test.py:110-175.  Greedy suppression in INPUT order: keep[i] = 1 iff box i overlaps no earlier kept box.
 *   mode 0 = separating-axis test of the bird's-eye rectangles (NMS_SAT; touching counts), mode 1 = 3-D IoU > iou_threshold with the
 *   kept box's centre nudged by 1e-4 (NMS_IOU).  boxes [n_max][7] fp32 (x, y, z, l, w, h, yaw); count_dev (may be NULL) = number of
 *   valid rows on the device; n_max <= 4096.  ws: dcf_eval_nms_workspace_bytes(n_max).
 * dcf_eval_match: test.py:177-206.  tp_counters[t] += number of predictions whose bird's-eye IoU with any labelled box
 *   (ref row [9], last column == 1) exceeds thresholds_dev[t] (fp64, on the device). */
int dcf_eval_score_filter(const float *pred, int B, int h, int w, float threshold, int cap, float *boxes_out, int32_t *count_out,
                          dcf_stream_t stream);
size_t dcf_eval_nms_workspace_bytes(int n_max);
int dcf_eval_nms(const float *boxes, const int32_t *count_dev, int n_max, int mode, double iou_threshold, int32_t *keep, int32_t *nkeep,
                 void *ws, dcf_stream_t stream);
int dcf_eval_match(const float *pred_boxes, int npred, const float *ref_boxes, int nref_rows, const double *thresholds_dev, int nthr,
                   int32_t *tp_counters, dcf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DCF_HIP_H */
