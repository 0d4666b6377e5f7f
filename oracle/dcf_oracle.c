/*
 * dcf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's per-frame
 * geometry path, used only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker for the HIP kernels.  Nothing under the
 * product package may link, load or call this file.
 *
 * Parity status
 *   voxelise (compat), range filter, projection+compaction: PINNED against the
 *     imported reference (oracle/gen_golden.py -> tests/golden/geometry_*.npz),
 *     bit-exact, generated under torch.use_deterministic_algorithms(True).
 *   voxelise (accum), KNN: the reference has no such code;
 *     "parity unpinned" -- this file IS the specification.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: every product/sum below is one fp32 rounding
 * exactly where the reference has one; the fused steps are explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------
 * Range filter.  Reference: data_import_carla.py:215-229.
 * keep iff  lo < p < hi  (strict) on all three axes, thresholds already
 * rounded to fp32 by the caller: hi = (float)(max - delta).  Order preserved.
 * lim = {xlo, xhi, ylo, yhi, zlo, zhi}.  Returns n_in; writes compacted points
 * and (optionally) their source row index.
 * ------------------------------------------------------------------------- */
int dcf_oracle_range_filter(const float *pts, int n, const float *lim,
                            float *out_pts, int32_t *out_src)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (x > lim[0] && x < lim[1] && y > lim[2] && y < lim[3] && z > lim[4] && z < lim[5]) {
            out_pts[3 * m] = x; out_pts[3 * m + 1] = y; out_pts[3 * m + 2] = z;
            if (out_src) out_src[m] = i;
            ++m;
        }
    }
    return m;
}

/* Index affine.  Reference: data_import_carla.py:35-43, :236 (matmul of
 * [x,y,z,1] with a 4x3 matrix whose only non-zeros are the scales and offsets).
 * Probed here: the CPU sgemm evaluates the K=4 dot product as the chain
 * fl(x*m0) -> fmaf(y,m1,.) -> fmaf(z,m2,.) -> fmaf(1,m3,.), which for this
 * sparse matrix is exactly fl(fl(p*scale) + offset). */
static inline float idx_affine(float p, float scale, float offset)
{
    float t = p * scale;
    return t + offset;
}

/* ---------------------------------------------------------------------------
 * Voxelise.  Reference: data_import_carla.py:236-258 (the "interpolate" branch).
 * aff = {sx, ox, sy, oy, sz, oz}.  grid is [Cz][L][W] fp32, zeroed here.
 * mode 0 = compat: eight sequential gather-add-scatter passes, last writer
 *          (highest point index) wins inside a pass -- what `V[idx] += w`
 *          does under deterministic algorithms.
 * mode 1 = accum : true trilinear splat, points added in index order.
 * mode 2 = occupancy: interpolate=False (:231-234): grid[zl][xl][yl] = 1.
 * ids (optional) receives trunc'd (x,y,z) voxel ids as [3][n] int64 (:258).
 * ------------------------------------------------------------------------- */
void dcf_oracle_voxelize(const float *pts, int n, const float *aff,
                         int Cz, int L, int W, int mode, float *grid, int64_t *ids)
{
    memset(grid, 0, sizeof(float) * (size_t)Cz * L * W);
    int64_t *vox = (int64_t *)malloc(sizeof(int64_t) * (size_t)n * 8);
    float *wgt = (float *)malloc(sizeof(float) * (size_t)n * 8);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        float fx = idx_affine(pts[3 * i], aff[0], aff[1]);
        float fy = idx_affine(pts[3 * i + 1], aff[2], aff[3]);
        float fz = idx_affine(pts[3 * i + 2], aff[4], aff[5]);
        int64_t xl = (int64_t)fx, yl = (int64_t)fy, zl = (int64_t)fz; /* trunc */
        int64_t xu = xl + 1, yu = yl + 1, zu = zl + 1;
        float dx = fx - (float)xl, dy = fy - (float)yl, dz = fz - (float)zl;
        float ax = 1.0f - dx, ay = 1.0f - dy, az = 1.0f - dz;
        if (ids) { ids[i] = xl; ids[(size_t)n + i] = yl; ids[2 * (size_t)n + i] = zl; }
        /* corner order and left-to-right products exactly as :250-257 */
        int64_t cz[8] = {zl, zu, zl, zu, zl, zu, zl, zu};
        int64_t cx[8] = {xl, xl, xu, xu, xl, xl, xu, xu};
        int64_t cy[8] = {yl, yl, yl, yl, yu, yu, yu, yu};
        float w[8];
        w[0] = (ax * ay) * az; w[1] = (ax * ay) * dz;
        w[2] = (dx * ay) * az; w[3] = (dx * ay) * dz;
        w[4] = (ax * dy) * az; w[5] = (ax * dy) * dz;
        w[6] = (dx * dy) * az; w[7] = (dx * dy) * dz;
        for (int c = 0; c < 8; ++c) {
            vox[(size_t)i * 8 + c] = (cz[c] * L + cx[c]) * W + cy[c];
            wgt[(size_t)i * 8 + c] = w[c];
        }
    }
    if (mode == 0) {
        for (int c = 0; c < 8; ++c) {
            for (int i = 0; i < n; ++i) tmp[i] = grid[vox[(size_t)i * 8 + c]] + wgt[(size_t)i * 8 + c];
            for (int i = 0; i < n; ++i) grid[vox[(size_t)i * 8 + c]] = tmp[i];
        }
    } else if (mode == 2) {
        for (int i = 0; i < n; ++i) grid[vox[(size_t)i * 8]] = 1.0f;
    } else {
        for (int i = 0; i < n; ++i)
            for (int c = 0; c < 8; ++c) grid[vox[(size_t)i * 8 + c]] += wgt[(size_t)i * 8 + c];
    }
    free(vox); free(wgt); free(tmp);
}

/* ---------------------------------------------------------------------------
 * Projection + in-image filter + compaction.
 * Reference: data_import_carla.py:196-210 (and :261-266 for the padding).
 * crt is the 4x3 row-major CRT_tensor (:34).  Dot products are the probed
 * sgemm chain: fl(x*c0) -> fmaf(y,c1,.) -> fmaf(z,c2,.) -> fmaf(1,c3,.).
 * mode 0 = compat: keep iff 0<u<ulim and 0<v<vlim where the reference passes
 *          ulim=image_height, vlim=image_width (sic, :202-205), no depth test.
 * mode 1 = correct: additionally require depth d > 0.
 * uv_out [n][2], xyz_out [n][3] receive the compacted survivors in order;
 * the caller zero-pads to max_num_pc.  Returns n_valid.
 * ------------------------------------------------------------------------- */
int dcf_oracle_project(const float *pts, int n, const float *crt, float ulim, float vlim,
                       int mode, float *uv_out, float *xyz_out, int32_t *src_out)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        float acc[3];
        for (int j = 0; j < 3; ++j) {
            float a = x * crt[0 * 3 + j];
            a = fmaf(y, crt[1 * 3 + j], a);
            a = fmaf(z, crt[2 * 3 + j], a);
            a = fmaf(1.0f, crt[3 * 3 + j], a);
            acc[j] = a;
        }
        float u = acc[0] / acc[2], v = acc[1] / acc[2];
        int keep = (u > 0.0f) && (u < ulim) && (v > 0.0f) && (v < vlim);
        if (mode == 1) keep = keep && (acc[2] > 0.0f);
        if (keep) {
            uv_out[2 * m] = u; uv_out[2 * m + 1] = v;
            xyz_out[3 * m] = x; xyz_out[3 * m + 1] = y; xyz_out[3 * m + 2] = z;
            if (src_out) src_out[m] = i;
            ++m;
        }
    }
    return m;
}

/* ---------------------------------------------------------------------------
 * BEV K-nearest-neighbour, brute force.  Specification: SURVEY.md App. D
 * (the reference has no KNN: model.py:199-203 is a TODO) -- parity unpinned.
 * Target pixel (i,j) of the stride-s site has metric centre
 *   X_i = ((i+0.5)*s - xo)/xs ,  Y_j = ((j+0.5)*s - yo)/ys        (one fp32 divide)
 * metric d2 = fl(fl(dx*dx) + fl(dy*dy)), dx = fl(x_k - X_i); total order
 * ascending (d2, k); output int32 [K][h][w], -1 where fewer than K candidates
 * or d2 > rmax2 (rmax2 < 0 means infinity).
 * ------------------------------------------------------------------------- */
void dcf_oracle_knn_bev(const float *xyz, int n, int K, int h, int w, int stride,
                        float xs, float xo, float ys, float yo, float rmax2, int32_t *out)
{
    /* pixel rows are independent: threaded over i when built with OpenMP (bench.py's cpu_baseline uses all host cores);
     * every pixel's scan is the same sequential loop either way, so the result does not depend on the thread count */
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        float *bd = (float *)malloc(sizeof(float) * (size_t)K);
        int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int i = 0; i < h; ++i) {
            float X = (((float)i + 0.5f) * (float)stride - xo) / xs;
            for (int j = 0; j < w; ++j) {
                float Y = (((float)j + 0.5f) * (float)stride - yo) / ys;
                int cnt = 0;
                for (int k = 0; k < n; ++k) {
                    float dx = xyz[3 * k] - X, dy = xyz[3 * k + 1] - Y;
                    float a = dx * dx, b = dy * dy;
                    float d2 = a + b;
                    if (rmax2 >= 0.0f && d2 > rmax2) continue;
                    /* insert keeping (d2, index) ascending; k ascends, so ties keep the earlier */
                    if (cnt < K || d2 < bd[cnt - 1]) {
                        int pos = (cnt < K) ? cnt : K - 1;
                        while (pos > 0 && bd[pos - 1] > d2) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                        bd[pos] = d2; bi[pos] = k;
                        if (cnt < K) ++cnt;
                    }
                }
                for (int q = 0; q < K; ++q)
                    out[((size_t)q * h + i) * w + j] = (q < cnt) ? bi[q] : -1;
            }
        }
        free(bd); free(bi);
    }
}

/* Same contract, for a LIST of pixels (pi[q], pj[q]) of the site instead of the whole
 * site: lets the tests sample a full-size site at random without a whole-site brute force.
 * out int32 [npix][K]. */
void dcf_oracle_knn_pixels(const float *xyz, int n, int K, const int32_t *pi, const int32_t *pj, int npix, int stride,
                           float xs, float xo, float ys, float yo, float rmax2, int32_t *out)
{
    float *bd = (float *)malloc(sizeof(float) * (size_t)K);
    int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
    for (int q = 0; q < npix; ++q) {
        float X = (((float)pi[q] + 0.5f) * (float)stride - xo) / xs;
        float Y = (((float)pj[q] + 0.5f) * (float)stride - yo) / ys;
        int cnt = 0;
        for (int k = 0; k < n; ++k) {
            float dx = xyz[3 * k] - X, dy = xyz[3 * k + 1] - Y;
            float a = dx * dx, b = dy * dy;
            float d2 = a + b;
            if (rmax2 >= 0.0f && d2 > rmax2) continue;
            if (cnt < K || d2 < bd[cnt - 1]) {
                int pos = (cnt < K) ? cnt : K - 1;
                while (pos > 0 && bd[pos - 1] > d2) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                bd[pos] = d2; bi[pos] = k;
                if (cnt < K) ++cnt;
            }
        }
        for (int t = 0; t < K; ++t) out[(size_t)q * K + t] = (t < cnt) ? bi[t] : -1;
    }
    free(bd); free(bi);
}
