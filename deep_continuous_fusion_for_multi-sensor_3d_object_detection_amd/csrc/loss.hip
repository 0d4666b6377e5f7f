// loss.hip -- the device half of the detection objective (loss.py:129-189 of the reference).
//
// Target assignment stays on the host (it consumes numpy's global RNG exactly like loss.py:74-127); what runs here is
// everything that touches the head outputs: the 2-way cross-entropy at the sampled positive / negative cells of both
// anchors (mean per list, loss.py:129-142), the Smooth-L1 of the encoded box offsets at the regression cells
// (loss.py:144-186) -- and their gradients, written straight into dense gradient maps.  One launch replaces the ~60 tiny
// gather / softmax / index_put kernels of the vectorised torch version.
#include "dcf_common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// the same for a 1024-thread workgroup (16 waves), in a fixed order
__device__ __forceinline__ float block_sum16(float v, float *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) t += (red[4 * q] + red[4 * q + 1]) + (red[4 * q + 2] + red[4 * q + 3]);
    return t;
}

// ints  = [B x {off_int, npos, nneg, nrow, off_float, nbox}] then per sample: pos cells, neg cells, reg cells, box of each reg cell
// floats = per sample: weight of each reg cell, then nbox x 7 box parameters
// (1024 threads: with the reference's 'last' reduction ONE workgroup does all the work, and its entries are chains of dependent
// loads -- list item -> cell -> scores -> atomics; at 256 threads the launch took 40 us between forward and backward)
__global__ void __launch_bounds__(1024) k_loss_fwd_bwd(const float *cls, int64_t cls_bs, const float *reg, int64_t reg_bs, const float *anc,
                                                      const int64_t *ints, const float *floats, int B, int HW, float gain, int reduction,
                                                      float *loss, float *gcls, int64_t gcls_bs, float *greg, int64_t greg_bs)
{
    __shared__ float red[16];
    const int b = blockIdx.x;
    // reduction 0 = 'last' (reference behaviour: only the last sample counts), 1 = 'sum', 2 = 'mean'
    if (reduction == 0 && b != B - 1) return;
    const float wsample = reduction == 2 ? 1.f / (float)B : 1.f;
    const int64_t *pl = ints + 6 * b;
    const int o = (int)pl[0], npos = (int)pl[1], nneg = (int)pl[2], nrow = (int)pl[3], of = (int)pl[4];
    const int64_t *pos = ints + o, *neg = pos + npos, *rows = neg + nneg, *rbox = rows + nrow;
    const float *wrow = floats + of, *boxes = wrow + nrow;
    const float *c = cls + b * cls_bs, *r = reg + b * reg_bs;
    float *gc = gcls + b * gcls_bs, *gr = greg + b * greg_bs;
    float acc = 0.f;
    // ---- classification: entry e = (anchor a, list item)
    const int ncls = 2 * (npos + nneg);
    for (int e = threadIdx.x; e < ncls; e += blockDim.x) {
        const int a = e / (npos + nneg), it = e - a * (npos + nneg);
        const bool is_pos = it < npos;
        const int cell = (int)(is_pos ? pos[it] : neg[it - npos]);
        const float inv = 1.f / (float)(is_pos ? npos : nneg);
        const float s0 = c[(int64_t)(2 * a) * HW + cell], s1 = c[(int64_t)(2 * a + 1) * HW + cell];
        const float m = fmaxf(s0, s1);
        const float e0 = expf(s0 - m), e1 = expf(s1 - m);
        const float lse = m + logf(e0 + e1);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        acc += (lse - (is_pos ? s1 : s0)) * inv;
        const float g = inv * wsample;
        atomicAdd(gc + (int64_t)(2 * a) * HW + cell, (p0 - (is_pos ? 0.f : 1.f)) * g);
        atomicAdd(gc + (int64_t)(2 * a + 1) * HW + cell, (p1 - (is_pos ? 1.f : 0.f)) * g);
    }
    // ---- regression: entry e = (row, anchor a, component j)
    float accr = 0.f;
    for (int e = threadIdx.x; e < nrow * 14; e += blockDim.x) {
        const int row = e / 14, q = e - row * 14;
        const int a = q / 7, j = q - a * 7;
        const int cell = (int)rows[row];
        const float *bx = boxes + (int64_t)rbox[row] * 7;
        const float *an = anc + (int64_t)a * 7 * HW + cell;     // an[j * HW]
        float t;
        if (j < 2) {
            const float l = an[3 * HW], w = an[4 * HW];
            t = (bx[j] - an[j * HW]) / sqrtf(l * l + w * w);
        } else if (j == 2) {
            t = (bx[2] - an[2 * HW]) / an[5 * HW];
        } else if (j < 6) {
            t = logf(bx[j] / an[j * HW]);
        } else {
            const float d = bx[6] - an[6 * HW];
            t = atan2f(sinf(d), cosf(d));
        }
        const float d = r[(int64_t)q * HW + cell] - t;
        const float ad = fabsf(d);
        accr += (ad < 1.f ? 0.5f * d * d : ad - 0.5f) * wrow[row];
        atomicAdd(gr + (int64_t)q * HW + cell, (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * wrow[row] * gain * wsample);
    }
    const float tot = block_sum16(acc + gain * accr, red);
    if (threadIdx.x == 0) atomicAdd(loss, tot * wsample);
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-side target assignment ("sampling: device"): loss.py:74-127 without the host -- the positive
// windows of the labelled boxes, a random subset of them when there are more than pos_cap, neg_count random cells that are
// not selected positives -- followed by the same loss terms as above, in ONE launch (one workgroup per sample).
//
// Randomness is a counter-based hash of (seed, sample, stream, index, attempt): no state, no order dependence, the same lists
// for the same seed on every launch -- tests/test_gpu_loss_sampling.py restates it in Python and compares the lists bit for
// bit.  Semantics kept from the reference: the positive list has one entry per (box, window cell) (overlapping windows give a
// cell twice), the subset is uniform without replacement over ENTRIES (loss.py:107-110: shuffle, truncate), negatives are
// drawn with replacement and rejected only against the selected positives (loss.py:117-126).
__host__ __device__ inline uint64_t dcf_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint32_t dcf_loss_rand(uint64_t seed, int sample, int stream, int index, int attempt)
{
    const uint64_t ctr = ((uint64_t)(uint32_t)sample << 44) | ((uint64_t)(uint32_t)stream << 40) | ((uint64_t)(uint32_t)attempt << 20) | (uint64_t)(uint32_t)index;
    return (uint32_t)(dcf_mix64(seed ^ dcf_mix64(ctr)) >> 32);
}

constexpr int LS_MAXE = 1024;       // positive entries per sample (max_box * span^2)
constexpr int LS_MAXNEG = 512;

struct LossSampleArgs {
    const float *cls, *reg, *anc, *boxes;
    const int32_t *nbox;
    int64_t cls_bs, reg_bs, gcls_bs, greg_bs;
    float *loss, *gcls, *greg;
    int32_t *pos_out, *neg_out, *counts_out;
    uint64_t seed;
    int max_box, box_stride, B, H, W, span, regress_type, pos_cap, neg_count, reduction;
    float xs, xo, ys, yo, rs, gain;
};

__global__ void __launch_bounds__(256) k_loss_sample_fwd_bwd(LossSampleArgs a)
{
    __shared__ float red[4];
    __shared__ int e_cell[LS_MAXE];            // entry -> cell (px * W + py)
    __shared__ short e_box[LS_MAXE];           // entry -> box
    __shared__ unsigned e_key[LS_MAXE];
    __shared__ int sel[LS_MAXE];               // selected positive cells (compacted, entry order)
    __shared__ int negs[LS_MAXNEG];
    __shared__ int box_first[65], box_cx[64], box_cy[64];
    __shared__ int s_np, s_nsel;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int HW = a.H * a.W, half = a.span / 2;
    const int nb = min(a.nbox[b], a.max_box);
    const float *bx = a.boxes + (int64_t)b * a.max_box * a.box_stride;
    // ---- positive entries: box k's window cells inside the map, in the reference's order (box, dx, dy)
    if (tid == 0) {
        int n = 0;
        for (int k = 0; k < nb; ++k) {
            // fp32 arithmetic and truncation as the reference does on 0-dim tensors (loss.py:85-86)
            const int cx = (int)(__fdiv_rn(__fadd_rn(__fmul_rn(bx[k * a.box_stride], a.xs), a.xo), a.rs));
            const int cy = (int)(__fdiv_rn(__fadd_rn(__fmul_rn(bx[k * a.box_stride + 1], a.ys), a.yo), a.rs));
            box_first[k] = n;
            box_cx[k] = cx; box_cy[k] = cy;
            if (cx < 0 || cx > a.H - 1 || cy < 0 || cy > a.W - 1) { box_cx[k] = -1000000; continue; }
            for (int dx = 0; dx < a.span; ++dx)
                for (int dy = 0; dy < a.span; ++dy) {
                    const int px = cx - half + dx, py = cy - half + dy;
                    if (px < 0 || px > a.H - 1 || py < 0 || py > a.W - 1) continue;
                    if (n < LS_MAXE) { e_cell[n] = px * a.W + py; e_box[n] = (short)k; }
                    ++n;
                }
        }
        box_first[nb] = n;
        s_np = min(n, LS_MAXE);
    }
    __syncthreads();
    const int np = s_np;
    // ---- subset of pos_cap entries when there are more: the pos_cap smallest (key, entry) pairs
    for (int i = tid; i < np; i += blockDim.x) e_key[i] = dcf_loss_rand(a.seed, b, 1, i, 0);
    __syncthreads();
    const bool cut = np > a.pos_cap;
    for (int i = tid; i < np; i += blockDim.x) {
        bool keep = true;
        if (cut) {
            int rank = 0;
            const unsigned ki = e_key[i];
            for (int j = 0; j < np; ++j) rank += (e_key[j] < ki || (e_key[j] == ki && j < i)) ? 1 : 0;
            keep = rank < a.pos_cap;
        }
        sel[i] = keep ? 1 : 0;                   // (flags first: the keys are still being read by other threads)
    }
    __syncthreads();
    if (tid == 0) {                              // order-preserving compaction in place (<= 1024 entries: a serial scan is microseconds)
        int n = 0;
        for (int i = 0; i < np; ++i)
            if (sel[i]) sel[n++] = e_cell[i];    // n <= i: the flag of entry i is read before slot n is written
        s_nsel = n;
    }
    __syncthreads();
    const int npos = s_nsel;
    // ---- negatives: item i keeps drawing until its cell is not a selected positive
    for (int i = tid; i < a.neg_count; i += blockDim.x) {
        int cell = 0;
        for (int att = 0; att < (1 << 20); ++att) {
            const uint32_t u = dcf_loss_rand(a.seed, b, 2, i, att);
            cell = (int)(((uint64_t)u * (uint64_t)HW) >> 32);
            bool hit = false;
            for (int j = 0; j < npos; ++j) hit |= sel[j] == cell;
            if (!hit) break;
        }
        negs[i] = cell;
    }
    __syncthreads();
    if (a.pos_out)
        for (int i = tid; i < a.pos_cap; i += blockDim.x) a.pos_out[(int64_t)b * a.pos_cap + i] = i < npos ? sel[i] : -1;
    if (a.neg_out)
        for (int i = tid; i < a.neg_count; i += blockDim.x) a.neg_out[(int64_t)b * a.neg_count + i] = negs[i];
    if (a.counts_out && tid == 0) { a.counts_out[2 * b] = npos; a.counts_out[2 * b + 1] = np; }
    // reduction 0 = 'last' (reference behaviour: only the last sample counts), 1 = 'sum', 2 = 'mean'
    if (a.reduction == 0 && b != a.B - 1) return;
    const float wsample = a.reduction == 2 ? 1.f / (float)a.B : 1.f;
    const float *c = a.cls + b * a.cls_bs, *r = a.reg + b * a.reg_bs;
    float *gc = a.gcls + b * a.gcls_bs, *gr = a.greg + b * a.greg_bs;
    const int nneg = a.neg_count;
    float acc = 0.f;
    const int ncls = 2 * (npos + nneg);
    for (int e = tid; e < ncls; e += blockDim.x) {
        const int an = e / (npos + nneg), it = e - an * (npos + nneg);
        const bool is_pos = it < npos;
        const int cell = is_pos ? sel[it] : negs[it - npos];
        const float inv = 1.f / (float)(is_pos ? npos : nneg);
        const float s0 = c[(int64_t)(2 * an) * HW + cell], s1 = c[(int64_t)(2 * an + 1) * HW + cell];
        const float m = fmaxf(s0, s1);
        const float e0 = expf(s0 - m), e1 = expf(s1 - m);
        const float lse = m + logf(e0 + e1);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        acc += (lse - (is_pos ? s1 : s0)) * inv;
        const float g = inv * wsample;
        atomicAdd(gc + (int64_t)(2 * an) * HW + cell, (p0 - (is_pos ? 0.f : 1.f)) * g);
        atomicAdd(gc + (int64_t)(2 * an + 1) * HW + cell, (p1 - (is_pos ? 1.f : 0.f)) * g);
    }
    // ---- regression rows: every window entry (regress_type 0) or the centre cell only; a box's rows share its weight 1 / (rows * 14)
    float accr = 0.f;
    for (int e = tid; e < np * 14; e += blockDim.x) {
        const int row = e / 14, q = e - row * 14;
        const int k = e_box[row], cell = e_cell[row];
        int nrows_k = box_first[k + 1] - box_first[k];
        if (a.regress_type != 0) {
            if (cell != box_cx[k] * a.W + box_cy[k]) continue;
            nrows_k = 1;
        }
        const float wrow = 1.f / (float)(nrows_k * 14);
        const int an_ = q / 7, j = q - an_ * 7;
        const float *bk = bx + (int64_t)k * a.box_stride;
        const float *an = a.anc + (int64_t)an_ * 7 * HW + cell;
        float t;
        if (j < 2) {
            const float l = an[3 * HW], w = an[4 * HW];
            t = (bk[j] - an[j * HW]) / sqrtf(l * l + w * w);
        } else if (j == 2) {
            t = (bk[2] - an[2 * HW]) / an[5 * HW];
        } else if (j < 6) {
            t = logf(bk[j] / an[j * HW]);
        } else {
            const float d = bk[6] - an[6 * HW];
            t = atan2f(sinf(d), cosf(d));
        }
        const float d = r[(int64_t)q * HW + cell] - t;
        const float ad = fabsf(d);
        accr += (ad < 1.f ? 0.5f * d * d : ad - 0.5f) * wrow;
        atomicAdd(gr + (int64_t)q * HW + cell, (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * wrow * a.gain * wsample);
    }
    const float tot = block_sum(acc + a.gain * accr, red);
    if (tid == 0) atomicAdd(a.loss, tot * wsample);
}

}  // namespace

extern "C" uint32_t dcf_loss_sample_rand(uint64_t seed, int sample, int stream, int index, int attempt)
{
    return dcf_loss_rand(seed, sample, stream, index, attempt);
}

extern "C" int dcf_loss_sample_fwd_bwd(const float *cls, int64_t cls_bstride, const float *reg, int64_t reg_bstride, const float *anchors,
                                       const float *boxes, const int32_t *nbox_dev, int max_box, int box_stride, int B, int H, int W,
                                       float xs, float xo, float ys, float yo, float reduced_scale, int span, int regress_type, int pos_cap,
                                       int neg_count, uint64_t seed, float reg_gain, int reduction, float *loss, float *gcls,
                                       int64_t gcls_bstride, float *greg, int64_t greg_bstride, int32_t *pos_out, int32_t *neg_out,
                                       int32_t *counts_out, dcf_stream_t stream)
{
    DCF_REQUIRE(cls && reg && anchors && boxes && nbox_dev && loss && gcls && greg && B > 0 && H > 0 && W > 0, "dcf_loss_sample_fwd_bwd: bad arguments");
    DCF_REQUIRE(reduction >= 0 && reduction <= 2, "dcf_loss_sample_fwd_bwd: reduction must be 0 (last), 1 (sum) or 2 (mean)");
    DCF_REQUIRE(max_box >= 0 && max_box <= 64 && span >= 1 && max_box * span * span <= LS_MAXE && box_stride >= 7,
                "dcf_loss_sample_fwd_bwd: at most 64 boxes and %d window cells per sample", LS_MAXE);
    DCF_REQUIRE(pos_cap >= 1 && pos_cap <= LS_MAXE && neg_count >= 1 && neg_count <= LS_MAXNEG, "dcf_loss_sample_fwd_bwd: pos_cap <= %d, neg_count <= %d", LS_MAXE, LS_MAXNEG);
    LossSampleArgs a;
    a.cls = cls; a.reg = reg; a.anc = anchors; a.boxes = boxes; a.nbox = nbox_dev;
    a.cls_bs = cls_bstride; a.reg_bs = reg_bstride; a.gcls_bs = gcls_bstride; a.greg_bs = greg_bstride;
    a.loss = loss; a.gcls = gcls; a.greg = greg; a.pos_out = pos_out; a.neg_out = neg_out; a.counts_out = counts_out;
    a.seed = seed; a.max_box = max_box; a.box_stride = box_stride; a.B = B; a.H = H; a.W = W; a.span = span;
    a.regress_type = regress_type; a.pos_cap = pos_cap; a.neg_count = neg_count; a.reduction = reduction;
    a.xs = xs; a.xo = xo; a.ys = ys; a.yo = yo; a.rs = reduced_scale; a.gain = reg_gain;
    hipStream_t s = S(stream);
    DCF_LAUNCH("loss_sample_fwd_bwd", s, hipLaunchKernelGGL(k_loss_sample_fwd_bwd, dim3(B), dim3(256), 0, s, a));
    return DCF_OK;
}

extern "C" int dcf_loss_fwd_bwd(const float *cls, int64_t cls_bstride, const float *reg, int64_t reg_bstride, const float *anchors,
                                const int64_t *ints, const float *floats, int B, int HW, float reg_gain, int reduction,
                                float *loss, float *gcls, int64_t gcls_bstride, float *greg, int64_t greg_bstride, dcf_stream_t stream)
{
    DCF_REQUIRE(cls && reg && anchors && ints && floats && loss && gcls && greg && B > 0 && HW > 0, "dcf_loss_fwd_bwd: bad arguments");
    DCF_REQUIRE(reduction >= 0 && reduction <= 2, "dcf_loss_fwd_bwd: reduction must be 0 (last), 1 (sum) or 2 (mean)");
    hipStream_t s = S(stream);
    DCF_LAUNCH("loss_fwd_bwd", s, hipLaunchKernelGGL(k_loss_fwd_bwd, dim3(B), dim3(1024), 0, s, cls, cls_bstride, reg, reg_bstride, anchors, ints,
                                                     floats, B, HW, reg_gain, reduction, loss, gcls, gcls_bstride, greg, greg_bstride));
    return DCF_OK;
}
