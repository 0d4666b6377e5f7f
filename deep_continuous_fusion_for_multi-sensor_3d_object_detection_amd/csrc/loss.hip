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

// ints  = [B x {off_int, npos, nneg, nrow, off_float, nbox}] then per sample: pos cells, neg cells, reg cells, box of each reg cell
// floats = per sample: weight of each reg cell, then nbox x 7 box parameters
__global__ void __launch_bounds__(256) k_loss_fwd_bwd(const float *cls, int64_t cls_bs, const float *reg, int64_t reg_bs, const float *anc,
                                                      const int64_t *ints, const float *floats, int B, int HW, float gain, int reduction,
                                                      float *loss, float *gcls, int64_t gcls_bs, float *greg, int64_t greg_bs)
{
    __shared__ float red[4];
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
    const float tot = block_sum(acc + gain * accr, red);
    if (threadIdx.x == 0) atomicAdd(loss, tot * wsample);
}

}  // namespace

extern "C" int dcf_loss_fwd_bwd(const float *cls, int64_t cls_bstride, const float *reg, int64_t reg_bstride, const float *anchors,
                                const int64_t *ints, const float *floats, int B, int HW, float reg_gain, int reduction,
                                float *loss, float *gcls, int64_t gcls_bstride, float *greg, int64_t greg_bstride, dcf_stream_t stream)
{
    DCF_REQUIRE(cls && reg && anchors && ints && floats && loss && gcls && greg && B > 0 && HW > 0, "dcf_loss_fwd_bwd: bad arguments");
    DCF_REQUIRE(reduction >= 0 && reduction <= 2, "dcf_loss_fwd_bwd: reduction must be 0 (last), 1 (sum) or 2 (mean)");
    hipStream_t s = S(stream);
    DCF_LAUNCH("loss_fwd_bwd", s, hipLaunchKernelGGL(k_loss_fwd_bwd, dim3(B), dim3(256), 0, s, cls, cls_bstride, reg, reg_bstride, anchors, ints,
                                                     floats, B, HW, reg_gain, reduction, loss, gcls, gcls_bstride, greg, greg_bstride));
    return DCF_OK;
}
