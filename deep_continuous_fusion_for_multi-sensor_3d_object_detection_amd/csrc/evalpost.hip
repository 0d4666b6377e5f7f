// evalpost.hip -- evaluation post-processing on the device: what /root/reference/test.py:88-206 does
// with Python loops over boxes on the host -- score threshold + compaction (:88-108), greedy rotated-box suppression in
// input order with the separating-axis test (:142-175) or the 3-D IoU (:110-140), and the bird's-eye-IoU matching behind
// the precision / recall counters (:177-206).  Geometry in fp64 like the host statement (evalgeom.py), decisions
// bit-for-bit the same on tests/golden/eval.npz (survivor indices, TP counters).
//
// Suppression is sequential by definition (a box survives iff it overlaps none of the EARLIER SURVIVORS); here the n^2
// pairwise tests run in parallel into a bit matrix, and one wave (lane = 64-box word of the survivor set) walks the rows.
// HBM traffic is negligible (n <= 4096 boxes): these kernels are latency-bound by design, not on the train-step path.
#include "dcf_common.h"

namespace {

constexpr int EV_CAP = 4096;            // boxes per sample the suppression handles (64 words x 64 bits)

struct EvGeom {                          // per box, computed once
    double rect[4][2];                   // SAT: bird's-eye rectangle in (x, y), size[0] along the heading
    double foot[4][2];                   // IoU: footprint in (x, z), corners 3,2,1,0 of the 8-corner box
    double footn[4][2];                  // the same for the centre nudged by +1e-4 (the survivor's role in NMS_IOU)
    double top, bot, topn, botn;         // y extent (corner 0 / corner 4), plain and nudged
    double area, vol, arean, voln;       // footprint area and box volume, plain and nudged (equal up to rounding)
};

__device__ void ev_corners(double cx, double cy, double cz, double l, double w, double h, double yaw, double (&X)[8], double (&Y)[8], double (&Z)[8])
{
    const double c = cos(yaw), s = sin(yaw);
    const double sgx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sgy[8] = {1, 1, 1, 1, -1, -1, -1, -1}, sgz[8] = {1, -1, -1, 1, 1, -1, -1, 1};
    for (int i = 0; i < 8; ++i) {
        const double sx = sgx[i] * (l / 2), sy = sgy[i] * (h / 2), sz = sgz[i] * (w / 2);
        X[i] = c * sx + s * sz + cx;
        Y[i] = sy + cy;
        Z[i] = -s * sx + c * sz + cz;
    }
}

__device__ double ev_shoelace(const double (*p)[2], int n)
{
    double a = 0.0, b = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + n - 1) % n;                  // roll(., 1)[i] = [i - 1]
        a += p[i][0] * p[j][1];
        b += p[i][1] * p[j][0];
    }
    return 0.5 * fabs(a - b);
}

// Intersection polygon of `subject` (4 vertices) with the convex polygon `clip` (4 vertices), half-plane by half-plane;
// strictly-left-of-edge is inside (evalgeom._clip_convex).  Returns the vertex count (<= 8).
__device__ int ev_clip(const double (*subject)[2], const double (*clip)[2], double (*out)[2])
{
    double cur[12][2], nxt[12][2];
    int n = 4;
    for (int i = 0; i < 4; ++i) { cur[i][0] = subject[i][0]; cur[i][1] = subject[i][1]; }
    double ax = clip[3][0], ay = clip[3][1];
    for (int e = 0; e < 4; ++e) {
        const double bx = clip[e][0], by = clip[e][1];
        if (n == 0) return 0;
        const double ex = bx - ax, ey = by - ay;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const int q = (i + n - 1) % n;
            const double sp = ex * (cur[i][1] - ay) - ey * (cur[i][0] - ax);
            const double sq = ex * (cur[q][1] - ay) - ey * (cur[q][0] - ax);
            if ((sp > 0) != (sq > 0)) {
                const double den = ex * (cur[i][1] - cur[q][1]) - ey * (cur[i][0] - cur[q][0]);
                const double t = (ex * (ay - cur[q][1]) - ey * (ax - cur[q][0])) / den;
                nxt[m][0] = cur[q][0] + t * (cur[i][0] - cur[q][0]);
                nxt[m][1] = cur[q][1] + t * (cur[i][1] - cur[q][1]);
                ++m;
            }
            if (sp > 0) { nxt[m][0] = cur[i][0]; nxt[m][1] = cur[i][1]; ++m; }
        }
        n = m;
        for (int i = 0; i < n; ++i) { cur[i][0] = nxt[i][0]; cur[i][1] = nxt[i][1]; }
        ax = bx; ay = by;
    }
    for (int i = 0; i < n; ++i) { out[i][0] = cur[i][0]; out[i][1] = cur[i][1]; }
    return n;
}

// (3-D IoU, bird's-eye IoU) of a candidate (subject) against a survivor / reference (clip): evalgeom.rotated_iou
__device__ void ev_iou(const double (*f1)[2], double top1, double bot1, double a1, double v1, const double (*f2)[2], double top2, double bot2,
                       double a2, double v2, double &iou3d, double &iou2d)
{
    double poly[12][2];
    const int n = ev_clip(f1, f2, poly);
    const double ia = n >= 3 ? ev_shoelace(poly, n) : 0.0;
    iou2d = ia / (a1 + a2 - ia);
    const double top = fmin(top1, top2), bot = fmax(bot1, bot2);
    const double iv = ia * fmax(0.0, top - bot);
    iou3d = iv / (v1 + v2 - iv);
}

__device__ bool ev_sat(const double (*A)[2], const double (*B)[2])
{
    for (int which = 0; which < 2; ++which) {
        const double (*P)[2] = which == 0 ? A : B;
        for (int i = 0; i < 4; ++i) {
            const int j = (i + 1) & 3;
            const double ex = P[j][0] - P[i][0], ey = P[j][1] - P[i][1];
            double nx = ey, ny = -ex;
            const double norm = hypot(nx, ny);
            nx /= norm; ny /= norm;
            double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
            for (int k = 0; k < 4; ++k) {
                const double pa = A[k][0] * nx + A[k][1] * ny, pb = B[k][0] * nx + B[k][1] * ny;
                amin = fmin(amin, pa); amax = fmax(amax, pa); bmin = fmin(bmin, pb); bmax = fmax(bmax, pb);
            }
            if (amax < bmin || bmax < amin) return false;
        }
    }
    return true;
}

__device__ void ev_prep_one(const float *b, EvGeom &g)
{
    const double cx = b[0], cy = b[1], cz = b[2], l = b[3], w = b[4], h = b[5], yaw = b[6];
    {   // evalgeom.bev_rect(c[:2], c[3:5], c[6])
        const double hl = l / 2, hw = w / 2, c = cos(yaw), s = sin(yaw);
        const double sl[4] = {-1, 1, 1, -1}, sw[4] = {-1, -1, 1, 1};
        for (int k = 0; k < 4; ++k) {
            g.rect[k][0] = cx + sl[k] * hl * c - sw[k] * hw * s;
            g.rect[k][1] = cy + sl[k] * hl * s + sw[k] * hw * c;
        }
    }
    double X[8], Y[8], Z[8];
    ev_corners(cx, cy, cz, l, w, h, yaw, X, Y, Z);
    const int order[4] = {3, 2, 1, 0};
    for (int k = 0; k < 4; ++k) { g.foot[k][0] = X[order[k]]; g.foot[k][1] = Z[order[k]]; }
    g.top = Y[0]; g.bot = Y[4];
    g.area = ev_shoelace(g.foot, 4);
    auto volume = [&]() {
        const double d01 = sqrt((X[0] - X[1]) * (X[0] - X[1]) + (Y[0] - Y[1]) * (Y[0] - Y[1]) + (Z[0] - Z[1]) * (Z[0] - Z[1]));
        const double d12 = sqrt((X[1] - X[2]) * (X[1] - X[2]) + (Y[1] - Y[2]) * (Y[1] - Y[2]) + (Z[1] - Z[2]) * (Z[1] - Z[2]));
        const double d04 = sqrt((X[0] - X[4]) * (X[0] - X[4]) + (Y[0] - Y[4]) * (Y[0] - Y[4]) + (Z[0] - Z[4]) * (Z[0] - Z[4]));
        return d01 * d12 * d04;
    };
    g.vol = volume();
    ev_corners(cx + 0.0001, cy + 0.0001, cz + 0.0001, l, w, h, yaw, X, Y, Z);
    for (int k = 0; k < 4; ++k) { g.footn[k][0] = X[order[k]]; g.footn[k][1] = Z[order[k]]; }
    g.topn = Y[0]; g.botn = Y[4];
    g.arean = ev_shoelace(g.footn, 4);
    g.voln = volume();
}

__global__ void __launch_bounds__(256) k_eval_prep(const float *boxes, const int *count, int n_max, EvGeom *geom)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = count ? min(*count, n_max) : n_max;
    if (i >= n) return;
    ev_prep_one(boxes + (size_t)i * 7, geom[i]);
}

// bit j of mask[i][j / 64] = box i (the candidate) overlaps box j < i (a possible survivor)
__global__ void __launch_bounds__(256) k_eval_pairs(const EvGeom *geom, const int *count, int n_max, int mode, double thr, int words, unsigned long long *mask)
{
    const int n = count ? min(*count, n_max) : n_max;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / words, wd = t - i * words;
    if (i >= n) return;
    unsigned long long bits = 0;
    const int j0 = wd * 64;
    if (j0 < i) {
        const EvGeom &gi = geom[i];
        for (int b = 0; b < 64; ++b) {
            const int j = j0 + b;
            if (j >= i) break;
            const EvGeom &gj = geom[j];
            bool over;
            if (mode == 0) {
                over = ev_sat(gi.rect, gj.rect);
            } else {
                double i3, i2;
                ev_iou(gi.foot, gi.top, gi.bot, gi.area, gi.vol, gj.footn, gj.topn, gj.botn, gj.arean, gj.voln, i3, i2);
                over = i3 > thr;
            }
            if (over) bits |= 1ull << b;
        }
    }
    mask[(size_t)i * words + wd] = bits;
}

// one wave: lane = word of the survivor bit set; rows are prefetched 8 ahead (their loads do not depend on the decisions)
__global__ void __launch_bounds__(64) k_eval_scan(const unsigned long long *mask, const int *count, int n_max, int words, int32_t *keep, int32_t *nkeep)
{
    const int n = count ? min(*count, n_max) : n_max;
    const int lane = threadIdx.x;
    unsigned long long kept = 0;
    int total = 0;
    constexpr int U = 8;
    for (int i0 = 0; i0 < n; i0 += U) {
        unsigned long long row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) row[u] = (i0 + u < n && lane < words) ? mask[(size_t)(i0 + u) * words + lane] : 0ull;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            if (i >= n) break;
            const bool hit = (row[u] & kept) != 0ull;
            const bool any = __ballot(hit) != 0ull;
            if (!any) {
                if (lane == (i >> 6)) kept |= 1ull << (i & 63);
                ++total;
            }
            if (lane == 0) keep[i] = any ? 0 : 1;
        }
    }
    if (lane == 0) *nkeep = total;
}

// one thread per kept prediction: OR over the labelled boxes of (bird's-eye IoU > t) per threshold (test.py:190-203)
__global__ void __launch_bounds__(256) k_eval_match(const float *pred, int npred, const float *refs, int nref_rows, const double *thr, int nthr, int32_t *tp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npred) return;
    EvGeom gp;
    ev_prep_one(pred + (size_t)i * 7, gp);
    unsigned flags = 0;
    for (int r = 0; r < nref_rows; ++r) {
        const float *rb = refs + (size_t)r * 9;
        if (rb[8] != 1.0f) continue;
        EvGeom gr;
        ev_prep_one(rb, gr);
        double i3, i2;
        ev_iou(gp.foot, gp.top, gp.bot, gp.area, gp.vol, gr.foot, gr.top, gr.bot, gr.area, gr.vol, i3, i2);
        for (int t = 0; t < nthr; ++t)
            if (i2 > thr[t]) flags |= 1u << t;
    }
    for (int t = 0; t < nthr; ++t)
        if (flags & (1u << t)) atomicAdd(&tp[t], 1);
}

// Score threshold + order-preserving compaction of one sample per workgroup (test.py:88-108): anchor 0's pixels in raster
// order, then anchor 1's.  pred [B][32][hw] fp32: class scores in channels 2a+1, decoded boxes in channels 18+7a .. +6.
__global__ void __launch_bounds__(1024) k_eval_score_filter(const float *pred, int hw, float thr, int cap, float *boxes, int32_t *count)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int b = blockIdx.x;
    const float *p = pred + (size_t)b * 32 * hw;
    float *out = boxes + (size_t)b * cap * 7;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c0 = 0; c0 < 2 * hw; c0 += 1024) {
        const int c = c0 + threadIdx.x;
        const int a = c >= hw ? 1 : 0, px = c - a * hw;
        const bool ok = c < 2 * hw && p[(size_t)(2 * a + 1) * hw + px] > thr;
        const unsigned long long bal = __ballot(ok);
        const int before = __popcll(bal & ((1ull << lane) - 1));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const int pos = off + before;
        if (ok && pos < cap)
            for (int k = 0; k < 7; ++k) out[(size_t)pos * 7 + k] = p[(size_t)(18 + 7 * a + k) * hw + px];
        __syncthreads();
        if (threadIdx.x == 0) { int tot = 0; for (int w = 0; w < 16; ++w) tot += wsum[w]; base_s += tot; }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[b] = base_s;          // may exceed cap: the caller checks
}

}  // namespace

extern "C" int dcf_eval_score_filter(const float *pred, int B, int h, int w, float thr, int cap, float *boxes, int32_t *count, dcf_stream_t stream)
{
    DCF_REQUIRE(pred && boxes && count && B > 0 && h > 0 && w > 0 && cap > 0, "dcf_eval_score_filter: bad arguments");
    hipStream_t s = S(stream);
    DCF_LAUNCH_B("eval_score_filter", (double)B * h * w * 2 * 4.0, s, hipLaunchKernelGGL(k_eval_score_filter, dim3(B), dim3(1024), 0, s, pred, h * w, thr, cap, boxes, count));
    return DCF_OK;
}

extern "C" size_t dcf_eval_nms_workspace_bytes(int n_max)
{
    const size_t words = (size_t)(n_max + 63) / 64;
    return sizeof(EvGeom) * (size_t)n_max + 8 * words * (size_t)n_max + 64;
}

extern "C" int dcf_eval_nms(const float *boxes, const int32_t *count_dev, int n_max, int mode, double iou_threshold, int32_t *keep, int32_t *nkeep,
                            void *ws, dcf_stream_t stream)
{
    DCF_REQUIRE(boxes && keep && nkeep && ws && n_max >= 0 && n_max <= EV_CAP, "dcf_eval_nms: at most %d boxes per call (got %d)", EV_CAP, n_max);
    DCF_REQUIRE(mode == 0 || mode == 1, "dcf_eval_nms: mode 0 = separating axes, 1 = 3-D IoU");
    hipStream_t s = S(stream);
    if (n_max == 0) { DCF_HIP(hipMemsetAsync(nkeep, 0, sizeof(int32_t), s)); return DCF_OK; }
    const int words = (n_max + 63) / 64;
    EvGeom *geom = (EvGeom *)ws;
    unsigned long long *mask = (unsigned long long *)((char *)ws + ((sizeof(EvGeom) * (size_t)n_max + 63) & ~(size_t)63));
    DCF_LAUNCH("eval_prep", s, hipLaunchKernelGGL(k_eval_prep, dim3(cdiv(n_max, 256)), dim3(256), 0, s, boxes, count_dev, n_max, geom));
    DCF_LAUNCH("eval_pairs", s, hipLaunchKernelGGL(k_eval_pairs, dim3(cdiv((int64_t)n_max * words, 256)), dim3(256), 0, s, geom, count_dev, n_max, mode, iou_threshold, words, mask));
    DCF_LAUNCH("eval_scan", s, hipLaunchKernelGGL(k_eval_scan, dim3(1), dim3(64), 0, s, mask, count_dev, n_max, words, keep, nkeep));
    return DCF_OK;
}

extern "C" int dcf_eval_match(const float *pred_boxes, int npred, const float *ref_boxes, int nref_rows, const double *thresholds_dev, int nthr,
                              int32_t *tp_counters, dcf_stream_t stream)
{
    DCF_REQUIRE(ref_boxes && thresholds_dev && tp_counters && nthr >= 1 && nthr <= 32 && npred >= 0 && nref_rows >= 0, "dcf_eval_match: bad arguments");
    if (npred == 0) return DCF_OK;
    DCF_REQUIRE(pred_boxes, "dcf_eval_match: null predictions");
    hipStream_t s = S(stream);
    DCF_LAUNCH("eval_match", s, hipLaunchKernelGGL(k_eval_match, dim3(cdiv(npred, 256)), dim3(256), 0, s, pred_boxes, npred, ref_boxes, nref_rows, thresholds_dev, nthr, tp_counters));
    return DCF_OK;
}
