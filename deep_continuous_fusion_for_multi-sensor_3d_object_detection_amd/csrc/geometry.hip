// geometry.hip -- per-frame geometry kernels: range filter, trilinear voxeliser,
// pinhole projection + compaction, BEV K-nearest-neighbour.
//
// Replaces CarlaDataset.Voxelization_Projection / Projection
// (/root/reference/data_import_carla.py:196-267), which run on the CPU inside the
// DataLoader.  All of it is HBM/latency bound integer+fp32 work: one thread per
// point / per BEV pixel, coalesced SoA-free reads of the [n][3] cloud, atomics only
// where the algorithm has a real collision (voxel ownership, cell histogram).
//
// Bit-exactness contract (checked against oracle/dcf_oracle.c): this file is
// compiled with -ffp-contract=off and spells every fused step as __fmaf_rn, so each
// product/sum rounds exactly where the reference's CPU arithmetic rounds.
#include <stdlib.h>
#include "dcf_common.h"

namespace {

struct Lim6 { float v[6]; };
struct Aff6 { float v[6]; };
struct Crt12 { float v[12]; };

__device__ __forceinline__ bool in_range(float x, float y, float z, const Lim6 &l)
{
    return x > l.v[0] && x < l.v[1] && y > l.v[2] && y < l.v[3] && z > l.v[4] && z < l.v[5];
}

// ------------------------------------------------------------------------------
// Order-preserving compaction: count -> scan -> scatter.  1024 items per block.
// ------------------------------------------------------------------------------
constexpr int CP_THREADS = 256;
constexpr int CP_ITEMS = 4;
constexpr int CP_TILE = CP_THREADS * CP_ITEMS;

// block-wide exclusive scan of one int per thread (256 threads = 4 waves); returns
// the exclusive prefix and writes the block total to *total (valid in all threads).
__device__ __forceinline__ int block_excl_scan(int v, int *total)
{
    __shared__ int wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int s = wsum[i];
        if (i < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

struct RangePred {
    Lim6 lim;
    __device__ bool operator()(const float *pts, int i, float *u, float *v) const
    {
        return in_range(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], lim);
    }
};

// data_import_carla.py:196-205 on top of the range filter of :215-226
struct ProjPred {
    Lim6 lim;
    Crt12 c;
    float ulim, vlim;
    int mode;
    __device__ bool operator()(const float *pts, int i, float *u, float *v) const
    {
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        if (!in_range(x, y, z, lim)) return false;
        float a[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float t = __fmul_rn(x, c.v[j]);       // the reference's K=4 sgemm chain
            t = __fmaf_rn(y, c.v[3 + j], t);
            t = __fmaf_rn(z, c.v[6 + j], t);
            t = __fmaf_rn(1.0f, c.v[9 + j], t);
            a[j] = t;
        }
        const float uu = __fdiv_rn(a[0], a[2]), vv = __fdiv_rn(a[1], a[2]);
        *u = uu;
        *v = vv;
        bool keep = uu > 0.0f && uu < ulim && vv > 0.0f && vv < vlim;
        if (mode == DCF_PROJ_CORRECT) keep = keep && a[2] > 0.0f;
        return keep;
    }
};

template <class Pred>
__global__ void __launch_bounds__(CP_THREADS) k_compact_count(const float *pts, int n, Pred pred, int *blocksum)
{
    const int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    int c = 0;
    float u, v;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        int i = base + k;
        if (i < n && pred(pts, i, &u, &v)) ++c;
    }
    int tot;
    block_excl_scan(c, &tot);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// single block: exclusive scan of nb block sums in place; total -> *count
__global__ void __launch_bounds__(CP_THREADS) k_compact_scan(int *blocksum, int nb, int *count, int fstride = 0)
{
    blocksum += (size_t)blockIdx.y * fstride; count += (size_t)blockIdx.y * fstride;
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += CP_THREADS) {
        int i = b0 + threadIdx.x;
        int v = i < nb ? blocksum[i] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot);
        if (i < nb) blocksum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *count = carry;
}

template <class Pred, bool EMIT_UV>
__global__ void __launch_bounds__(CP_THREADS) k_compact_scatter(const float *pts, int n, Pred pred, const int *blockoff,
                                                                float *out_uv, float *out_xyz, int *out_src)
{
    const int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    bool keep[CP_ITEMS];
    float u[CP_ITEMS], v[CP_ITEMS];
    int c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        int i = base + k;
        keep[k] = (i < n) && pred(pts, i, &u[k], &v[k]);
        c += keep[k] ? 1 : 0;
    }
    int tot;
    int pos = blockoff[blockIdx.x] + block_excl_scan(c, &tot);
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        if (keep[k]) {
            int i = base + k;
            out_xyz[3 * pos] = pts[3 * i];
            out_xyz[3 * pos + 1] = pts[3 * i + 1];
            out_xyz[3 * pos + 2] = pts[3 * i + 2];
            if (EMIT_UV) {
                out_uv[2 * pos] = u[k];
                out_uv[2 * pos + 1] = v[k];
            }
            if (out_src) out_src[pos] = i;
            ++pos;
        }
    }
}

// ---- the frames of a batch in one launch per phase (round 5: dcf_project_filter_batch): blockIdx.y = frame, every frame with its
// own points, count and projection matrix (KITTI calibrates per frame); the bodies are the per-frame kernels' own.
#define DCF_PROJ_BATCH_MAX 8
struct ProjBatch {
    const float *pts[DCF_PROJ_BATCH_MAX];
    int n[DCF_PROJ_BATCH_MAX];
    Crt12 c[DCF_PROJ_BATCH_MAX];
};
__device__ __forceinline__ ProjPred proj_pred_of(const ProjBatch &pb, int b, const Lim6 &lim, float ulim, float vlim, int mode)
{
    ProjPred p;
    p.lim = lim; p.ulim = ulim; p.vlim = vlim; p.mode = mode;
#pragma unroll
    for (int f = 0; f < DCF_PROJ_BATCH_MAX; ++f)             // (static indices: a run-time index into the by-value argument would go through scratch)
        if (f == b)
#pragma unroll
            for (int k = 0; k < 12; ++k) p.c.v[k] = pb.c[f].v[k];
    return p;
}
__device__ __forceinline__ const float *proj_pts_of(const ProjBatch &pb, int b, int &n)
{
    const float *q = nullptr;
    n = 0;
#pragma unroll
    for (int f = 0; f < DCF_PROJ_BATCH_MAX; ++f)
        if (f == b) { q = pb.pts[f]; n = pb.n[f]; }
    return q;
}

__global__ void __launch_bounds__(CP_THREADS) k_proj_count_b(ProjBatch pb, Lim6 lim, float ulim, float vlim, int mode, int *blocksum, int bstride)
{
    const int b = blockIdx.y;
    int n;
    const float *pts = proj_pts_of(pb, b, n);
    const ProjPred pred = proj_pred_of(pb, b, lim, ulim, vlim, mode);
    const int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    int c = 0;
    float u, v;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        const int i = base + k;
        if (i < n && pred(pts, i, &u, &v)) ++c;
    }
    int tot;
    block_excl_scan(c, &tot);
    if (threadIdx.x == 0) blocksum[(size_t)b * bstride + blockIdx.x] = tot;
}

// one block per frame: exclusive scan of the frame's nb block sums in place; total -> count[frame]
__global__ void __launch_bounds__(CP_THREADS) k_proj_scan_b(int *blocksum, int nb, int bstride, int *count)
{
    blocksum += (size_t)blockIdx.y * bstride;
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += CP_THREADS) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? blocksum[i] : 0;
        int tot;
        const int ex = block_excl_scan(v, &tot);
        if (i < nb) blocksum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) count[blockIdx.y] = carry;
}

__global__ void __launch_bounds__(CP_THREADS) k_proj_scatter_b(ProjBatch pb, Lim6 lim, float ulim, float vlim, int mode, const int *blockoff, int bstride,
                                                               float *out_uv, float *out_xyz, int rows)
{
    const int b = blockIdx.y;
    int n;
    const float *pts = proj_pts_of(pb, b, n);
    const ProjPred pred = proj_pred_of(pb, b, lim, ulim, vlim, mode);
    out_uv += (size_t)b * rows * 2; out_xyz += (size_t)b * rows * 3;
    const int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    bool keep[CP_ITEMS];
    float u[CP_ITEMS], v[CP_ITEMS];
    int c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        const int i = base + k;
        keep[k] = (i < n) && pred(pts, i, &u[k], &v[k]);
        c += keep[k] ? 1 : 0;
    }
    int tot;
    int pos = blockoff[(size_t)b * bstride + blockIdx.x] + block_excl_scan(c, &tot);
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        if (keep[k]) {
            const int i = base + k;
            out_xyz[3 * pos] = pts[3 * i];
            out_xyz[3 * pos + 1] = pts[3 * i + 1];
            out_xyz[3 * pos + 2] = pts[3 * i + 2];
            out_uv[2 * pos] = u[k];
            out_uv[2 * pos + 1] = v[k];
            ++pos;
        }
    }
}

// ------------------------------------------------------------------------------
// Voxeliser.  data_import_carla.py:236-258.
// ------------------------------------------------------------------------------
struct Corner8 {
    int vox[8];
    float w[8];
};

// index affine = fl(fl(p*scale)+offset)  (the sgemm chain on the sparse 4x3 matrix)
__device__ __forceinline__ void corners(float x, float y, float z, const Aff6 &a, int L, int W, Corner8 &o)
{
    const float fx = __fadd_rn(__fmul_rn(x, a.v[0]), a.v[1]);
    const float fy = __fadd_rn(__fmul_rn(y, a.v[2]), a.v[3]);
    const float fz = __fadd_rn(__fmul_rn(z, a.v[4]), a.v[5]);
    const int xl = (int)fx, yl = (int)fy, zl = (int)fz;  // trunc, like .type(torch.long)
    const float dx = __fsub_rn(fx, (float)xl), dy = __fsub_rn(fy, (float)yl), dz = __fsub_rn(fz, (float)zl);
    const float ax = __fsub_rn(1.0f, dx), ay = __fsub_rn(1.0f, dy), az = __fsub_rn(1.0f, dz);
    const float pxy[4] = {__fmul_rn(ax, ay), __fmul_rn(dx, ay), __fmul_rn(ax, dy), __fmul_rn(dx, dy)};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int zz = zl + (c & 1), xx = xl + ((c >> 1) & 1), yy = yl + ((c >> 2) & 1);
        o.vox[c] = (zz * L + xx) * W + yy;
        o.w[c] = __fmul_rn(pxy[c >> 1], (c & 1) ? dz : az);
    }
}

// Round r (0..8) of the compat voxeliser: resolve pass r-1, claim pass r.
// owner[parity][voxel] holds (highest claiming point index + 1), 0 = unclaimed.
__global__ void __launch_bounds__(256) k_voxel_compat_round(const float *pts, int n, Lim6 lim, Aff6 aff, int L, int W,
                                                            int nvox, int round, float *grid, int *owner)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (!in_range(x, y, z, lim)) return;
    Corner8 c8;
    corners(x, y, z, aff, L, W, c8);
    if (round >= 1) {
        const int c = round - 1;
        int *ow = owner + (size_t)(c & 1) * nvox;
        const int v = c8.vox[c];
        if (ow[v] == i + 1) {  // this point is the last writer of pass c
            grid[v] = __fadd_rn(grid[v], c8.w[c]);
            ow[v] = 0;
        }
    }
    if (round <= 7) {
        int *ow = owner + (size_t)(round & 1) * nvox;
        atomicMax(&ow[c8.vox[round]], i + 1);
    }
}

// frames of a batch: blockIdx.y = frame
struct VoxBatch {
    const float *pts[DCF_MAX_VOXEL_BATCH];
    int n[DCF_MAX_VOXEL_BATCH];
};
// ---- compat voxeliser in three launches instead of nine rounds.
// Pass c of the reference writes voxel cell(i)+delta_c for every point i, last writer (highest index) wins: the points that
// compete for one voxel in one pass are exactly the points of one CELL, so a single "last point of the cell" map decides
// every pass.  A voxel v then holds  sum over c = 0..7, in pass order, of  w_c(last(v - delta_c))  (cells without a point
// contribute nothing).  Step 1 claims the cells (atomicMax of index+1), step 2 lets every cell winner evaluate that ordered
// sum for its 8 corners (neighbouring winners compute the same value for a shared voxel: identical stores), step 3 clears
// the claims so the workspace is all-zero again.
__device__ __forceinline__ float corner_weight(float x, float y, float z, const Aff6 &a, int c)
{
    const float fx = __fadd_rn(__fmul_rn(x, a.v[0]), a.v[1]);
    const float fy = __fadd_rn(__fmul_rn(y, a.v[2]), a.v[3]);
    const float fz = __fadd_rn(__fmul_rn(z, a.v[4]), a.v[5]);
    const float dx = __fsub_rn(fx, (float)(int)fx), dy = __fsub_rn(fy, (float)(int)fy), dz = __fsub_rn(fz, (float)(int)fz);
    const float wx = ((c >> 1) & 1) ? dx : __fsub_rn(1.0f, dx);
    const float wy = ((c >> 2) & 1) ? dy : __fsub_rn(1.0f, dy);
    const float wz = (c & 1) ? dz : __fsub_rn(1.0f, dz);
    return __fmul_rn(__fmul_rn(wx, wy), wz);
}

__global__ void __launch_bounds__(256) k_voxel_cell_claim(VoxBatch vb, Lim6 lim, Aff6 aff, int L, int W, int nvox, int *owners, int release)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vb.n[b]) return;
    const float *pts = vb.pts[b];
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (!in_range(x, y, z, lim)) return;
    const int xl = (int)__fadd_rn(__fmul_rn(x, aff.v[0]), aff.v[1]), yl = (int)__fadd_rn(__fmul_rn(y, aff.v[2]), aff.v[3]),
              zl = (int)__fadd_rn(__fmul_rn(z, aff.v[4]), aff.v[5]);
    int *last = owners + (size_t)b * 2 * nvox;
    const int cell = (zl * L + xl) * W + yl;
    if (release) last[cell] = 0;
    else atomicMax(&last[cell], i + 1);
}

// TO / NHWC: fp32 [Cz][L][W] grids (the reference's layout) or the engine's input image [L][W][Cz] in the compute type --
// the value of a voxel is produced in one piece here, so rounding it on the way out equals casting the fp32 grid later.
template <typename TO, bool NHWC>
__global__ void __launch_bounds__(256) k_voxel_cell_gather(VoxBatch vb, Lim6 lim, Aff6 aff, int Cz, int L, int W, int nvox, TO *grids,
                                                           const int *owners)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= vb.n[b]) return;
    const float *pts = vb.pts[b];
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (!in_range(x, y, z, lim)) return;
    const int xl = (int)__fadd_rn(__fmul_rn(x, aff.v[0]), aff.v[1]), yl = (int)__fadd_rn(__fmul_rn(y, aff.v[2]), aff.v[3]),
              zl = (int)__fadd_rn(__fmul_rn(z, aff.v[4]), aff.v[5]);
    const int *last = owners + (size_t)b * 2 * nvox;
    if (last[(zl * L + xl) * W + yl] != i + 1) return;              // not the last point of its cell
    TO *grid = grids + (size_t)b * nvox;
    // contrib[k][c]: what pass c adds to this cell's corner k.  The contributor of (k, c) is the last point of the cell at
    // offset e_k - e_c (e = the corner's (z, x, y) bits), so the 27 neighbouring cells are visited once each -- one claim
    // read, one point read, the fractional parts computed once -- and hand their weights to the corners they share.
    // Absent contributors leave +0.0f, which an ordered sum of non-negative terms absorbs exactly.
    float contrib[8][8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) contrib[k][c] = 0.f;
    // (everything in FLAT voxel indices, like the reference's index arithmetic on the flattened grid: a corner index one
    // past the end of a row is the first voxel of the next row -- only reachable with limits that leave less than one
    // cell of margin, but the nine-round formulation and the CPU restatement behave that way)
    const int cell = (zl * L + xl) * W + yl;
#pragma unroll
    for (int oz = -1; oz <= 1; ++oz)
#pragma unroll
        for (int ox = -1; ox <= 1; ++ox)
#pragma unroll
            for (int oy = -1; oy <= 1; ++oy) {
                const int nc = cell + (oz * L + ox) * W + oy;
                if (nc < 0 || nc >= nvox) continue;
                const int j1 = last[nc];
                if (j1 <= 0) continue;
                const float *q = pts + 3 * (size_t)(j1 - 1);
                const float fx = __fadd_rn(__fmul_rn(q[0], aff.v[0]), aff.v[1]);
                const float fy = __fadd_rn(__fmul_rn(q[1], aff.v[2]), aff.v[3]);
                const float fz = __fadd_rn(__fmul_rn(q[2], aff.v[4]), aff.v[5]);
                const float dx = __fsub_rn(fx, (float)(int)fx), dy = __fsub_rn(fy, (float)(int)fy), dz = __fsub_rn(fz, (float)(int)fz);
                const float wxs[2] = {__fsub_rn(1.0f, dx), dx}, wys[2] = {__fsub_rn(1.0f, dy), dy}, wzs[2] = {__fsub_rn(1.0f, dz), dz};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ez = (k & 1) - oz, ex = ((k >> 1) & 1) - ox, ey = ((k >> 2) & 1) - oy;     // e_c = e_k - offset
                    if (ez < 0 || ez > 1 || ex < 0 || ex > 1 || ey < 0 || ey > 1) continue;
                    contrib[k][ez | (ex << 1) | (ey << 2)] = __fmul_rn(__fmul_rn(wxs[ex], wys[ey]), wzs[ez]);
                }
            }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) sum = __fadd_rn(sum, contrib[k][c]);                       // pass order
        const int v = cell + ((k & 1) * L + ((k >> 1) & 1)) * W + ((k >> 2) & 1);
        if (v >= nvox) continue;
        if (NHWC) {
            const int vz = v / (L * W), r = v - vz * (L * W);
            DT<TO>::st(grid + (size_t)r * Cz + vz, sum);                                       // r = vx*W + vy
        } else {
            DT<TO>::st(grid + v, sum);
        }
    }
}

__global__ void __launch_bounds__(256) k_voxel_accum(const float *pts, int n, Lim6 lim, Aff6 aff, int L, int W, float *grid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (!in_range(x, y, z, lim)) return;
    Corner8 c8;
    corners(x, y, z, aff, L, W, c8);
#pragma unroll
    for (int c = 0; c < 8; ++c) atomicAdd(&grid[c8.vox[c]], c8.w[c]);
}

// interpolate=False (data_import_carla.py:231-234): the voxel holding the point (trunc'd ids = lower corner) is set to 1;
// every writer stores the same value, so the result does not depend on the order
__global__ void __launch_bounds__(256) k_voxel_occupancy(const float *pts, int n, Lim6 lim, Aff6 aff, int L, int W, float *grid)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (!in_range(x, y, z, lim)) return;
    Corner8 c8;
    corners(x, y, z, aff, L, W, c8);
    grid[c8.vox[0]] = 1.0f;
}

// ------------------------------------------------------------------------------
// BEV KNN.  Points are counting-sorted into the stride-s BEV cells (8x8-blocked cell
// order so that a coarse block's points are one contiguous range), then every BEV
// pixel searches fine rings first and coarse block rings when the neighbourhood is
// sparse.  Result = K smallest by (d2, index) -- independent of visiting order.
// ------------------------------------------------------------------------------
struct KnnGrid {
    int h, w, h8, w8, stride;
    float xs, xo, ys, yo;
    // batched launches (grid.y = frame): per-frame strides of the point rows (floats), the valid counts (ints), the workspace
    // (ints) and the index maps (ints); all 0 for a single frame
    int fs_xyz, fs_cnt, fs_ws, fs_out;
};

__device__ __forceinline__ int cell_key(int ci, int cj, const KnnGrid &g)
{
    return (((ci >> 3) * g.w8 + (cj >> 3)) << 6) + ((ci & 7) << 3) + (cj & 7);
}

__device__ __forceinline__ void point_cell(float x, float y, const KnnGrid &g, int &ci, int &cj)
{
    const float fx = __fadd_rn(__fmul_rn(x, g.xs), g.xo), fy = __fadd_rn(__fmul_rn(y, g.ys), g.yo);
    ci = (int)floorf(fx / (float)g.stride);
    cj = (int)floorf(fy / (float)g.stride);
    ci = min(max(ci, 0), g.h - 1);
    cj = min(max(cj, 0), g.w - 1);
}

__global__ void __launch_bounds__(256) k_knn_hist(const float *xyz, const int *count, int n_max, KnnGrid g, int *cellcnt,
                                                  int *pkey)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    xyz += (size_t)blockIdx.y * g.fs_xyz; count += blockIdx.y * g.fs_cnt; cellcnt += (size_t)blockIdx.y * g.fs_ws; pkey += (size_t)blockIdx.y * g.fs_ws;
    const int n = min(*count, n_max);
    if (i >= n) return;
    int ci, cj;
    point_cell(xyz[3 * i], xyz[3 * i + 1], g, ci, cj);
    const int key = cell_key(ci, cj, g);
    pkey[i] = key;
    atomicAdd(&cellcnt[key], 1);
}

// generic multi-block exclusive scan over ints (count -> start), 3 phases
__global__ void __launch_bounds__(CP_THREADS) k_scan_blocksum(const int *in, int n, int *blocksum, int fstride = 0)
{
    in += (size_t)blockIdx.y * fstride; blocksum += (size_t)blockIdx.y * fstride;
    const int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    int c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k)
        if (base + k < n) c += in[base + k];
    int tot;
    block_excl_scan(c, &tot);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(CP_THREADS) k_scan_apply(const int *in, int n, const int *blockoff, int *out, int *cursor, int fstride = 0)
{
    in += (size_t)blockIdx.y * fstride; blockoff += (size_t)blockIdx.y * fstride; out += (size_t)blockIdx.y * fstride; cursor += (size_t)blockIdx.y * fstride;
    const int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    int v[CP_ITEMS];
    int c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        c += v[k];
    }
    int tot;
    int pos = blockoff[blockIdx.x] + block_excl_scan(c, &tot);
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        if (base + k < n) {
            out[base + k] = pos;
            cursor[base + k] = pos;
            pos += v[k];
        }
    }
}

__global__ void __launch_bounds__(256) k_knn_fill(const float *xyz, const int *count, int n_max, const int *pkey, int *cursor,
                                                  float4 *sorted, int fs_xyz = 0, int fs_cnt = 0, int fs_ws = 0)
{
    xyz += (size_t)blockIdx.y * fs_xyz; count += blockIdx.y * fs_cnt; pkey += (size_t)blockIdx.y * fs_ws; cursor += (size_t)blockIdx.y * fs_ws;
    sorted = reinterpret_cast<float4 *>(reinterpret_cast<int *>(sorted) + (size_t)blockIdx.y * fs_ws);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(*count, n_max);
    if (i >= n) return;
    const int pos = atomicAdd(&cursor[pkey[i]], 1);
    sorted[pos] = make_float4(xyz[3 * i], xyz[3 * i + 1], __int_as_float(i), 0.f);
}

// ---- the cell sort of SEVERAL sites of one batch in one launch per phase (dcf_knn_bev_sites): blockIdx.y = site * B + frame.
// Same arithmetic as the single-site kernels above on each site's own workspace (cellcnt | cellstart | cursor | blocksum | pkey |
// sorted), so the searches that follow see bit-identical tables.
#define DCF_MAX_KNN_SITES 4
struct KnnSortSites {
    KnnGrid g[DCF_MAX_KNN_SITES];
    int *ws[DCF_MAX_KNN_SITES];            // frame 0 of the site's workspace (frames g.fs_ws ints apart)
    int nscan[DCF_MAX_KNN_SITES], nsb[DCF_MAX_KNN_SITES];
    int pkey_off[DCF_MAX_KNN_SITES], sorted_off[DCF_MAX_KNN_SITES];    // in ints from the workspace start
    int n, B;
};

__global__ void __launch_bounds__(256) k_knn_zero_ms(KnnSortSites S)
{
    const int site = blockIdx.y / S.B, frame = blockIdx.y - site * S.B;
    int *p = S.ws[site] + (size_t)frame * S.g[site].fs_ws;
    const int n = S.nscan[site], n4 = n >> 2;
    int4 *row = reinterpret_cast<int4 *>(p);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) row[i] = make_int4(0, 0, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = 0;
}

__global__ void __launch_bounds__(256) k_knn_hist_ms(const float *xyz, const int *count, int n_max, KnnSortSites S)
{
    const int site = blockIdx.y / S.B, frame = blockIdx.y - site * S.B;
    const KnnGrid &g = S.g[site];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    xyz += (size_t)frame * g.fs_xyz; count += frame * g.fs_cnt;
    int *cellcnt = S.ws[site] + (size_t)frame * g.fs_ws, *pkey = cellcnt + S.pkey_off[site];
    const int n = min(*count, n_max);
    if (i >= n) return;
    int ci, cj;
    point_cell(xyz[3 * i], xyz[3 * i + 1], g, ci, cj);
    const int key = cell_key(ci, cj, g);
    pkey[i] = key;
    atomicAdd(&cellcnt[key], 1);
}

__global__ void __launch_bounds__(CP_THREADS) k_scan_blocksum_ms(KnnSortSites S)
{
    const int site = blockIdx.y / S.B, frame = blockIdx.y - site * S.B;
    if ((int)blockIdx.x >= S.nsb[site]) return;
    const int n = S.nscan[site];
    const int *in = S.ws[site] + (size_t)frame * S.g[site].fs_ws;
    int *blocksum = const_cast<int *>(in) + 3 * n;
    const int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    int c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k)
        if (base + k < n) c += in[base + k];
    int tot;
    block_excl_scan(c, &tot);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(CP_THREADS) k_compact_scan_ms(KnnSortSites S)
{
    const int site = blockIdx.y / S.B, frame = blockIdx.y - site * S.B;
    const int nb = S.nsb[site];
    int *blocksum = S.ws[site] + (size_t)frame * S.g[site].fs_ws + 3 * S.nscan[site];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += CP_THREADS) {
        int i = b0 + threadIdx.x;
        int v = i < nb ? blocksum[i] : 0;
        int tot;
        int ex = block_excl_scan(v, &tot);
        if (i < nb) blocksum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) blocksum[nb] = carry;
}

__global__ void __launch_bounds__(CP_THREADS) k_scan_apply_ms(KnnSortSites S)
{
    const int site = blockIdx.y / S.B, frame = blockIdx.y - site * S.B;
    if ((int)blockIdx.x >= S.nsb[site]) return;
    const int n = S.nscan[site];
    const int *in = S.ws[site] + (size_t)frame * S.g[site].fs_ws;
    int *out = const_cast<int *>(in) + n, *cursor = out + n;
    const int *blockoff = cursor + n;
    const int base = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS;
    int v[CP_ITEMS];
    int c = 0;
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        c += v[k];
    }
    int tot;
    int pos = blockoff[blockIdx.x] + block_excl_scan(c, &tot);
#pragma unroll
    for (int k = 0; k < CP_ITEMS; ++k) {
        if (base + k < n) {
            out[base + k] = pos;
            cursor[base + k] = pos;
            pos += v[k];
        }
    }
}

__global__ void __launch_bounds__(256) k_knn_fill_ms(const float *xyz, const int *count, int n_max, KnnSortSites S)
{
    const int site = blockIdx.y / S.B, frame = blockIdx.y - site * S.B;
    const KnnGrid &g = S.g[site];
    xyz += (size_t)frame * g.fs_xyz; count += frame * g.fs_cnt;
    int *ws = S.ws[site] + (size_t)frame * g.fs_ws;
    const int *pkey = ws + S.pkey_off[site];
    int *cursor = ws + 2 * S.nscan[site];
    float4 *sorted = reinterpret_cast<float4 *>(ws + S.sorted_off[site]);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = min(*count, n_max);
    if (i >= n) return;
    const int pos = atomicAdd(&cursor[pkey[i]], 1);
    sorted[pos] = make_float4(xyz[3 * i], xyz[3 * i + 1], __int_as_float(i), 0.f);
}

// K best (d2, index) pairs in ascending lexicographic order.  Round 4: insertion is a branch-free compare-exchange chain on
// 64-bit keys (bits of d2 above the index: for the non-negative d2 of a distance the unsigned order of the bits IS the float
// order, so the key order is the (d2, index) order of rounds 1-3; a NaN or infinite d2 sorts behind the empty slot's 3.0e38 and is
// never kept, as before).  The branchy version cost ~540 cycles per candidate in k_knn_search (tools/knn_stamps.py): exec-mask
// juggling through scalar registers on a wave that is alone on its SIMD.
template <int K>
struct TopK {
    float d[K];
    int id[K];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int q = 0; q < K; ++q) { d[q] = 3.0e38f; id[q] = 0x7fffffff; }
    }
    __device__ __forceinline__ float kth() const { return d[K - 1]; }
    __device__ __forceinline__ bool full() const { return id[K - 1] != 0x7fffffff; }
    __device__ __forceinline__ int count() const
    {
        int c = 0;
#pragma unroll
        for (int q = 0; q < K; ++q) c += id[q] != 0x7fffffff;
        return c;
    }
    static __device__ __forceinline__ unsigned long long key(float dd, int ii)
    {
        return ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)ii;
    }
    __device__ __forceinline__ void insert(float dd, int ii)
    {
        unsigned long long x = key(dd, ii);
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const unsigned long long kq = key(d[q], id[q]);
            const bool lt = x < kq;
            const unsigned long long lo = lt ? x : kq;
            x = lt ? kq : x;
            d[q] = __uint_as_float((unsigned)(lo >> 32));
            id[q] = (int)(unsigned)lo;
        }
    }
    // a key of another list that may already be in this one (merging lists that share earlier merges): a second copy is dropped
    __device__ __forceinline__ void insert_unique(unsigned long long x)
    {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const unsigned long long kq = key(d[q], id[q]);
            if (x == kq) x = ~0ull;
            const bool lt = x < kq;
            const unsigned long long lo = lt ? x : kq;
            x = lt ? kq : x;
            d[q] = __uint_as_float((unsigned)(lo >> 32));
            id[q] = (int)(unsigned)lo;
        }
    }
};

// Per-tile counters of k_knn_search (tools/knn_stamps.py; -DKNN_STAMP builds only): cycles and candidate points of the window
// phase and of the ring phase, rings walked, blocks scanned -- which tiles the launch waits for, and why.
#ifdef KNN_STAMP
#define KNN_STAMP_TILES 4096
__device__ long long g_knn_stamps[KNN_STAMP_TILES][16];
#define KNN_ST(k, v) do { if (blockIdx.y == 0 && tile < KNN_STAMP_TILES && lane == 0 && wv == 0) g_knn_stamps[tile][k] = (v); } while (0)
#define KNN_CNT(var, n) (var) += (n)
#define KNN_T(k) KNN_ST(k, __builtin_amdgcn_s_memtime() - st_t0)
#else
#define KNN_ST(k, v) do { } while (0)
#define KNN_CNT(var, n) do { } while (0)
#define KNN_T(k) do { } while (0)
#endif

// One workgroup per 8x8 tile of BEV pixels (= one coarse block of the cell grid), lane = pixel: all lanes walk the SAME cells /
// coarse rings, so control flow is uniform and every candidate point is a register broadcast; only the K-best insertion is per lane.
//   phase A: the 12x12 cell window around the tile (every lane's fine rings 0..2 are inside it)
//   phase B: coarse block rings around the tile for lanes whose neighbourhood is sparse
// A lane stops as soon as every unvisited point is provably farther than its K-th candidate.
// Round 4: FOUR waves per tile (one wave per tile before).  A tile's time is one serial chain -- candidate after candidate through
// the lanes' K-best lists, ~30 dependent instructions each -- and the launch waits for its longest chain with one or two waves on
// every SIMD (tools/knn_stamps.py).  The four waves walk the same cells and rings and each takes every fourth candidate; their
// lists are merged through LDS where a decision needs the pixel's true K-th (end of the window phase, end of a ring, after a
// block of more than KNN_MERGE_POINTS points), after which all four hold the same list and decide alike.
constexpr int KNN_MERGE_POINTS = 256;
template <int K, int NW>
__device__ __forceinline__ void knn_search_tile(const int *count, int n_max, const KnnGrid &g, const int *cellstart,
                                                const float4 *sorted, float rmax2, int *out, const int tile, const int wv)
{
    count += blockIdx.y * g.fs_cnt; cellstart += (size_t)blockIdx.y * g.fs_ws; out += (size_t)blockIdx.y * g.fs_out;
    sorted = reinterpret_cast<const float4 *>(reinterpret_cast<const int *>(sorted) + (size_t)blockIdx.y * g.fs_ws);
    const int lane = threadIdx.x & 63;
    const int TI = tile / g.w8, TJ = tile - TI * g.w8;
    const int i = TI * 8 + (lane >> 3), j = TJ * 8 + (lane & 7);
    const bool inside = (i < g.h) && (j < g.w);
    __shared__ unsigned long long s_keys[NW][K][64];
    const float s = (float)g.stride;
    const float X = __fdiv_rn(__fsub_rn(__fmul_rn((float)i + 0.5f, s), g.xo), g.xs);
    const float Y = __fdiv_rn(__fsub_rn(__fmul_rn((float)j + 0.5f, s), g.yo), g.ys);
    const float cwx = s / g.xs, cwy = s / g.ys;
    const float cwmin = fminf(cwx, cwy);
    const int n = min(*count, n_max);

    TopK<K> top;
    top.clear();
    bool done = !inside;
#ifdef KNN_STAMP
    long long st_pts = 0, st_rings = 0, st_blocks = 0;
    const long long st_t0 = __builtin_amdgcn_s_memtime();
#endif

    // candidates reach the lanes as register broadcasts: the wave loads up to 64 points with ONE vector load and then
    // hands them round with v_readlane (a dependent scalar load per point leaves the wave waiting on memory per point)
    // (four candidates per trip: their distance arithmetic is independent, only the compare-exchange chains are serial; lanes
    // beyond n hold NaN coordinates, which no test accepts)
    const float d2lim = rmax2 >= 0.0f ? rmax2 : __int_as_float(0x7f800000);
    const float4 nopoint = make_float4(__int_as_float(0x7fc00000), __int_as_float(0x7fc00000), 0.f, 0.f);
    auto offer = [&](const float4 &mine, int n, bool take) {
        const float lim = take ? d2lim : -1.0f;                 // a lane that does not take accepts no distance
        for (int u0 = wv; u0 < n; u0 += 4 * NW) {               // this wave's candidates: wv, wv + NW, wv + 2 NW, ... (< 64)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int u = u0 + NW * v;
                const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.x), u));
                const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.y), u));
                const int qi = __builtin_amdgcn_readlane(__float_as_int(mine.z), u);
                const float dx = __fsub_rn(qx, X), dy = __fsub_rn(qy, Y);
                const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
                top.insert(d2 <= lim ? d2 : __int_as_float(0x7f800000), qi);
            }
        }
    };
    // the four waves' lists of every pixel merged into each of them (uniform: every wave of the tile comes here together)
    auto merge = [&]() {
        if constexpr (NW > 1) {
#pragma unroll
            for (int q = 0; q < K; ++q) s_keys[wv][q][lane] = TopK<K>::key(top.d[q], top.id[q]);
            __syncthreads();
#pragma unroll
            for (int o = 1; o < NW; ++o) {
#pragma unroll
                for (int q = 0; q < K; ++q) top.insert_unique(s_keys[(wv + o) % NW][q][lane]);
            }
            __syncthreads();
        }
    };
    auto scan_range = [&](int b, int e, bool take) {
        KNN_CNT(st_pts, e - b);
        for (int p0 = b; p0 < e; p0 += 64) {
            const int n = min(64, e - p0);
            const float4 mine = lane < n ? sorted[p0 + lane] : nopoint;
            offer(mine, n, take);
        }
    };

    KNN_T(8);                                                  // the frame's point count has arrived
    if (n > 0) {
        // ---- phase A: rows 8TI-2 .. 8TI+9, three column segments (one per coarse block column)
        const int H8 = g.h8 * 8, W8 = g.w8 * 8;
        {
            // the 12 rows x 3 segments are looked up one per lane; the wave then scans the non-empty ranges
            const int row = lane / 3, seg = lane - row * 3;
            const int ci = TI * 8 - 2 + row;
            int c0 = TJ * 8 + (seg == 0 ? -2 : (seg == 1 ? 0 : 8));
            int c1 = TJ * 8 + (seg == 0 ? -1 : (seg == 1 ? 7 : 9));
            c0 = max(c0, 0); c1 = min(c1, W8 - 1);
            int ps = 0, pe = 0;
            if (lane < 36 && ci >= 0 && ci <= H8 - 1 && c0 <= c1) {
                ps = cellstart[cell_key(ci, c0, g)];
                pe = cellstart[cell_key(ci, c1, g) + 1];
            }
            // segmented gather: the (short) ranges are concatenated -- slot t of the concatenation belongs to the range r with
            // start_r <= t < start_r + len_r -- so the whole window arrives with one vector load per 64 points
            const int len = pe - ps;
            int inc = len;
#ifdef KNN_STAMP
            if (__any(len > 1 << 30)) return;
            KNN_T(9);                                           // the window's cell ranges have arrived
#endif
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(inc, o, 64);
                if (lane >= o) inc += v;
            }
            const int total = __builtin_amdgcn_readlane(inc, 63);
            const int exc = inc - len;
            KNN_ST(2, (long long)total);
            for (int base = 0; base < total; base += 64) {
                // slot t belongs to the first range whose inclusive prefix exceeds t: a six-step binary search over the lanes'
                // prefixes (empty ranges repeat their predecessor's prefix and are never found).  (A loop over the 36 ranges with
                // scalar broadcasts took 5.6 k of a median tile's 38 k cycles.)
                const int t = base + lane;
                int lo = 0, hi = 63;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int mid = (lo + hi) >> 1;
                    const bool right = __shfl(inc, mid, 64) <= t;
                    lo = right ? mid + 1 : lo;
                    hi = right ? hi : mid;
                }
                const int src = __shfl(ps, lo, 64) + (t - __shfl(exc, lo, 64));
                const int n = min(64, total - base);
                const float4 mine = lane < n ? sorted[src] : nopoint;
#ifdef KNN_STAMP
                if (base == 0) {
                    KNN_T(10);                                  // gather addresses computed
                    if (__any(mine.z == 123456.f)) return;
                    KNN_T(11);                                  // first 64 points have arrived
                }
#endif
                offer(mine, n, !done);
            }
        }
        merge();
        {
            const float bound = 2.5f * cwmin - 1e-3f;         // every unvisited point is >= 2.5 cells away
            if (!done && top.full() && top.kth() < bound * bound) done = true;
            if (rmax2 >= 0.0f && bound > 0.f && bound * bound > rmax2) done = true;
        }
        KNN_ST(0, __builtin_amdgcn_s_memtime() - st_t0);
#ifdef KNN_STAMP
        { const long long left = __popcll(__ballot(!done)); KNN_ST(6, left); }
#endif
        // ---- phase B: coarse rings (lanes still searching restart; order of visits does not matter)
        if (!__all(done)) {
            if (!done) top.clear();
            const int Rmax = max(g.h8, g.w8);
            for (int R = 0; R <= Rmax; ++R) {
                // the 8R blocks of ring R are probed 64 at a time, one per lane (tiles far from every point walk many
                // empty rings: a vector load per 64 blocks instead of two dependent scalar loads per block); the
                // non-empty ones are then scanned by the whole wave, in any order
                const int nblk = R == 0 ? 1 : 8 * R;
                for (int t0 = 0; t0 < nblk; t0 += 64) {
                    const int t = t0 + lane;
                    int a = 0, b = 0;
                    if (R > 0) {
                        if (t < 2 * R + 1) { a = -R; b = -R + t; }
                        else if (t < 4 * R + 2) { a = R; b = -R + (t - (2 * R + 1)); }
                        else if (t < 6 * R + 1) { b = -R; a = -R + 1 + (t - (4 * R + 2)); }
                        else { b = R; a = -R + 1 + (t - (6 * R + 1)); }
                    }
                    const int bi = TI + a, bj = TJ + b;
                    int ps = 0, pe = 0;
                    if (t < nblk && bi >= 0 && bi < g.h8 && bj >= 0 && bj < g.w8) {
                        const int k0 = (bi * g.w8 + bj) << 6;
                        ps = cellstart[k0];
                        pe = cellstart[k0 + 64];
                    }
                    unsigned long long live = __ballot(pe > ps);
                    while (live) {
                        const int l = __ffsll((long long)live) - 1;
                        live &= live - 1;
                        const int cbi = __builtin_amdgcn_readlane(bi, l), cbj = __builtin_amdgcn_readlane(bj, l);
                        const int cps = __builtin_amdgcn_readlane(ps, l), cpe = __builtin_amdgcn_readlane(pe, l);
                        bool visit = !done;
                        if (visit && top.full()) {  // prune by the block's metric bounding box
                            const float lox = ((float)(cbi * 8) * s - g.xo) / g.xs, hix = ((float)(cbi * 8 + 8) * s - g.xo) / g.xs;
                            const float loy = ((float)(cbj * 8) * s - g.yo) / g.ys, hiy = ((float)(cbj * 8 + 8) * s - g.yo) / g.ys;
                            const float ddx = fmaxf(0.f, fmaxf(lox - X, X - hix) - 1e-3f);
                            const float ddy = fmaxf(0.f, fmaxf(loy - Y, Y - hiy) - 1e-3f);
                            visit = (ddx * ddx + ddy * ddy) <= top.kth();
                        }
                        if (__any(visit)) { KNN_CNT(st_blocks, 1); scan_range(cps, cpe, visit); }
                        if (cpe - cps > KNN_MERGE_POINTS) merge();      // (the next blocks are pruned with the true K-th)
                    }
                }
                merge();
                KNN_CNT(st_rings, 1);
                const float bound = (8.0f * (float)R + 0.5f) * cwmin - 1e-3f;
                if (!done && top.full() && top.kth() < bound * bound) done = true;
                if (rmax2 >= 0.0f && bound > 0.f && bound * bound > rmax2) done = true;
                if (__all(done)) break;
            }
        }
    }
#ifdef KNN_STAMP
    KNN_ST(1, __builtin_amdgcn_s_memtime() - st_t0);
    KNN_ST(3, st_pts); KNN_ST(4, st_rings); KNN_ST(5, st_blocks);
#endif
    if (inside) {
        const int hw = g.h * g.w, pix = i * g.w + j;
#pragma unroll
        for (int q = 0; q < K; ++q) out[q * hw + pix] = top.id[q] != 0x7fffffff ? top.id[q] : -1;
    }
}

template <int K, int NW>
__global__ void __launch_bounds__(64 * NW) k_knn_search(const int *count, int n_max, KnnGrid g, const int *cellstart,
                                                       const float4 *sorted, float rmax2, int *out)
{
    knn_search_tile<K, NW>(count, n_max, g, cellstart, sorted, rmax2, out, blockIdx.x, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// Coarse sites (few pixels, many points per cell): one WAVE per pixel, lanes split the candidate
// points of every visited cell range, each keeping a private K-best; the wave-wide K-best is then
// extracted with K rounds of a 64-lane lexicographic (d2, index) min.  Same traversal and the same
// termination bounds as the tile kernel, so the result is the same exact K-best.
template <int K>
__device__ __forceinline__ void wave_merge(const TopK<K> &mine, float (&gd)[K], int (&gi)[K], int &gcnt)
{
    TopK<K> t = mine;  // popped copy
    gcnt = 0;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        float d = t.d[0];
        int id = t.id[0];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float od = __shfl_xor(d, o, 64);
            const int oi = __shfl_xor(id, o, 64);
            if (od < d || (od == d && oi < id)) { d = od; id = oi; }
        }
        gd[q] = d; gi[q] = id;
        if (id != 0x7fffffff) ++gcnt;
        if (t.id[0] == id && id != 0x7fffffff) {   // the winning lane pops its head (ids are unique)
#pragma unroll
            for (int k = 0; k + 1 < K; ++k) { t.d[k] = t.d[k + 1]; t.id[k] = t.id[k + 1]; }
            t.d[K - 1] = 3.0e38f; t.id[K - 1] = 0x7fffffff;
        }
    }
}

template <int K>
__device__ __forceinline__ void knn_search_wave(const int *count, int n_max, const KnnGrid &g, const int *cellstart,
                                                const float4 *sorted, float rmax2, int *out, const int bx)
{
    count += blockIdx.y * g.fs_cnt; cellstart += (size_t)blockIdx.y * g.fs_ws; out += (size_t)blockIdx.y * g.fs_out;
    sorted = reinterpret_cast<const float4 *>(reinterpret_cast<const int *>(sorted) + (size_t)blockIdx.y * g.fs_ws);
    const int lane = threadIdx.x & 63;
    const int pix = __builtin_amdgcn_readfirstlane(bx * 4 + (threadIdx.x >> 6));
    const int hw = g.h * g.w;
    if (pix >= hw) return;
    const int i = pix / g.w, j = pix - i * g.w;
    const float s = (float)g.stride;
    const float X = __fdiv_rn(__fsub_rn(__fmul_rn((float)i + 0.5f, s), g.xo), g.xs);
    const float Y = __fdiv_rn(__fsub_rn(__fmul_rn((float)j + 0.5f, s), g.yo), g.ys);
    const float cwmin = fminf(s / g.xs, s / g.ys);
    const int n = min(*count, n_max);

    TopK<K> top;
    top.clear();
    float gd[K];
    int gi[K];
    int gcnt = 0;
#pragma unroll
    for (int q = 0; q < K; ++q) { gd[q] = 3.0e38f; gi[q] = 0x7fffffff; }

    auto scan_range = [&](int b, int e) {
        for (int p = b + lane; p < e; p += 64) {
            const float4 q = sorted[p];
            const float dx = __fsub_rn(q.x, X), dy = __fsub_rn(q.y, Y);
            const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            if (!(rmax2 >= 0.0f && d2 > rmax2)) top.insert(d2, __float_as_int(q.z));
        }
    };

    bool done = (n == 0);
    if (!done) {
        const int H8 = g.h8 * 8, W8 = g.w8 * 8;
        {
            // 5 rows x (columns j-2..j+2 split at a coarse-block boundary: cells of one block row are contiguous keys):
            // at most 10 ranges, looked up one per lane, the non-empty ones scanned by the wave
            const int row = lane >> 1, part = lane & 1;
            const int ci = i - 2 + row;
            const int cbeg = max(j - 2, 0), cend = min(j + 2, W8 - 1);
            const int split = min(cend, (cbeg | 7));                  // last column of the first block
            const int c0 = part == 0 ? cbeg : split + 1;
            const int c1 = part == 0 ? split : cend;
            int ps = 0, pe = 0;
            if (lane < 10 && ci >= 0 && ci <= H8 - 1 && c0 <= c1) {
                ps = cellstart[cell_key(ci, c0, g)];
                pe = cellstart[cell_key(ci, c1, g) + 1];
            }
            unsigned long long live = __ballot(pe > ps);
            while (live) {
                const int l = __ffsll((long long)live) - 1;
                live &= live - 1;
                scan_range(__builtin_amdgcn_readlane(ps, l), __builtin_amdgcn_readlane(pe, l));
            }
        }
        wave_merge<K>(top, gd, gi, gcnt);
        const float bound = 2.5f * cwmin - 1e-3f;
        if (gcnt >= K && gd[K - 1] < bound * bound) done = true;
        if (rmax2 >= 0.0f && bound > 0.f && bound * bound > rmax2) done = true;
    }
    if (!done) {
        top.clear();
        gcnt = 0;
#pragma unroll
        for (int q = 0; q < K; ++q) { gd[q] = 3.0e38f; gi[q] = 0x7fffffff; }
        const int I = i >> 3, J = j >> 3;
        const int Rmax = max(g.h8, g.w8);
        for (int R = 0; R <= Rmax && !done; ++R) {
            // ring R's blocks probed one per lane, the non-empty ones scanned by the wave (as in k_knn_search)
            const int nblk = R == 0 ? 1 : 8 * R;
            for (int t0 = 0; t0 < nblk; t0 += 64) {
                const int t = t0 + lane;
                int a = 0, b = 0;
                if (R > 0) {
                    if (t < 2 * R + 1) { a = -R; b = -R + t; }
                    else if (t < 4 * R + 2) { a = R; b = -R + (t - (2 * R + 1)); }
                    else if (t < 6 * R + 1) { b = -R; a = -R + 1 + (t - (4 * R + 2)); }
                    else { b = R; a = -R + 1 + (t - (6 * R + 1)); }
                }
                const int bi = I + a, bj = J + b;
                int ps = 0, pe = 0;
                if (t < nblk && bi >= 0 && bi < g.h8 && bj >= 0 && bj < g.w8) {
                    const int k0 = (bi * g.w8 + bj) << 6;
                    ps = cellstart[k0];
                    pe = cellstart[k0 + 64];
                }
                unsigned long long live = __ballot(pe > ps);
                while (live) {
                    const int l = __ffsll((long long)live) - 1;
                    live &= live - 1;
                    const int cbi = __builtin_amdgcn_readlane(bi, l), cbj = __builtin_amdgcn_readlane(bj, l);
                    const int cps = __builtin_amdgcn_readlane(ps, l), cpe = __builtin_amdgcn_readlane(pe, l);
                    bool visit = true;
                    if (gcnt >= K) {   // prune with the (possibly stale, hence conservative) wave-wide K-th
                        const float lox = ((float)(cbi * 8) * s - g.xo) / g.xs, hix = ((float)(cbi * 8 + 8) * s - g.xo) / g.xs;
                        const float loy = ((float)(cbj * 8) * s - g.yo) / g.ys, hiy = ((float)(cbj * 8 + 8) * s - g.yo) / g.ys;
                        const float ddx = fmaxf(0.f, fmaxf(lox - X, X - hix) - 1e-3f);
                        const float ddy = fmaxf(0.f, fmaxf(loy - Y, Y - hiy) - 1e-3f);
                        visit = (ddx * ddx + ddy * ddy) <= gd[K - 1];
                    }
                    if (visit) scan_range(cps, cpe);
                }
            }
            wave_merge<K>(top, gd, gi, gcnt);
            const float bound = (8.0f * (float)R + 0.5f) * cwmin - 1e-3f;
            if (gcnt >= K && gd[K - 1] < bound * bound) done = true;
            if (rmax2 >= 0.0f && bound > 0.f && bound * bound > rmax2) done = true;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) out[q * hw + pix] = (q < gcnt) ? gi[q] : -1;
    }
}

template <int K>
__global__ void __launch_bounds__(256) k_knn_search_wave(const int *count, int n_max, KnnGrid g, const int *cellstart,
                                                         const float4 *sorted, float rmax2, int *out)
{
    knn_search_wave<K>(count, n_max, g, cellstart, sorted, rmax2, out, blockIdx.x);
}

// Coarse sites, dense regions on a FINER site's cells (round 3).  k_knn_search_wave looks at the site's own cells: at stride 8 / 16 a
// cell is 0.8 m / 1.6 m wide and the 5 x 5 window a wave scans first holds hundreds to thousands of points near the sensor.
// k_knn_search_fine runs the same search, one wave per pixel, in two phases on two cell structures:
//   phase 0: the 5 x 5 window of the site's own cells, as k_knn_search_wave -- unless it holds more than KNN_FINE_MIN_POINTS points:
//   phase A: (those dense pixels) square windows of the FINE site's cells (0.2 m; built anyway for the stride-2 site), radius
//            2, 4, 6, 8 cells; the cells of a window are probed one per lane and every lane inserts ITS cell's few points into
//            its private K-best; a 64-lane lexicographic merge and the termination test after each window.  Pixels with K
//            points within ~1.5 m finish here having touched a few dozen points;
//   phase B: the rest (sparse neighbourhoods, pixels outside the camera frustum) walk rings of the COARSE site's own 8 x 8-cell
//            blocks with bounding-box pruning, exactly as k_knn_search_wave does (on the fine grid the same pixels would walk
//            ~8x as many rings: measured 115 us per launch against 64).
// Termination of phase A is exact: the K-th best so far must be closer than the nearest OPEN edge of the scanned window (an edge
// on the grid border is closed: points beyond the grid are clamped into the border cells, which the window then contains).
// Same distance arithmetic, same (d2, index) order, same -1 padding as the other two kernels: bit-identical maps.
constexpr int KNN_FINE_MIN_POINTS = 192;      // points in a pixel's 5 x 5 window of own cells above which the fine cells are searched instead

__device__ __forceinline__ float open_edge_distance(const KnnGrid &gf, float X, float Y, int i0, int i1, int j0, int j1)
{
    const float s = (float)gf.stride;
    float b = 3.0e38f;
    if (i0 > 0) b = fminf(b, X - ((float)i0 * s - gf.xo) / gf.xs);
    if (i1 < gf.h - 1) b = fminf(b, ((float)(i1 + 1) * s - gf.xo) / gf.xs - X);
    if (j0 > 0) b = fminf(b, Y - ((float)j0 * s - gf.yo) / gf.ys);
    if (j1 < gf.w - 1) b = fminf(b, ((float)(j1 + 1) * s - gf.yo) / gf.ys - Y);
    return b - 1e-3f;                       // 1 mm margin for the fp32 rounding of the edges (as the other kernels)
}

template <int K>
__device__ __forceinline__ void knn_search_fine(const int *count, int n_max, const KnnGrid &g, const KnnGrid &gf, const int *cellstart_c,
                                                const float4 *sorted_c, const int *cellstart, const float4 *sorted, float rmax2,
                                                int *out, int dense_min, const int bx)
{
    // g: the coarse site (pixels, output, its own cells cellstart_c / sorted_c); gf: the fine site (cellstart / sorted)
    count += blockIdx.y * g.fs_cnt; out += (size_t)blockIdx.y * g.fs_out;
    cellstart_c += (size_t)blockIdx.y * g.fs_ws;
    sorted_c = reinterpret_cast<const float4 *>(reinterpret_cast<const int *>(sorted_c) + (size_t)blockIdx.y * g.fs_ws);
    cellstart += (size_t)blockIdx.y * gf.fs_ws;
    sorted = reinterpret_cast<const float4 *>(reinterpret_cast<const int *>(sorted) + (size_t)blockIdx.y * gf.fs_ws);
    const int lane = threadIdx.x & 63;
    const int pix = __builtin_amdgcn_readfirstlane(bx * 4 + (threadIdx.x >> 6));
    const int hw = g.h * g.w;
    if (pix >= hw) return;
    const int i = pix / g.w, j = pix - i * g.w;
    const float s = (float)g.stride;
    const float X = __fdiv_rn(__fsub_rn(__fmul_rn((float)i + 0.5f, s), g.xo), g.xs);
    const float Y = __fdiv_rn(__fsub_rn(__fmul_rn((float)j + 0.5f, s), g.yo), g.ys);
    const int n = min(*count, n_max);
    int ci, cj;
    point_cell(X, Y, gf, ci, cj);           // fine cell of the pixel centre (clamped into the grid)

    TopK<K> top;
    top.clear();
    float gd[K];
    int gi[K];
    int gcnt = 0;
#pragma unroll
    for (int q = 0; q < K; ++q) { gd[q] = 3.0e38f; gi[q] = 0x7fffffff; }
    auto scan_own = [&](int b, int e) {     // this LANE's cell
        for (int p = b; p < e; ++p) {
            const float4 q = sorted[p];
            const float dx = __fsub_rn(q.x, X), dy = __fsub_rn(q.y, Y);
            const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            if (!(rmax2 >= 0.0f && d2 > rmax2)) top.insert(d2, __float_as_int(q.z));
        }
    };
    auto scan_range = [&](int b, int e) {   // the wave together, on the coarse site's own cells
        for (int p = b + lane; p < e; p += 64) {
            const float4 q = sorted_c[p];
            const float dx = __fsub_rn(q.x, X), dy = __fsub_rn(q.y, Y);
            const float d2 = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
            if (!(rmax2 >= 0.0f && d2 > rmax2)) top.insert(d2, __float_as_int(q.z));
        }
    };
    bool done = (n == 0);
    // ---- phase 0: how many points does the 5 x 5 window of the site's OWN cells hold (the first thing k_knn_search_wave scans)?
    // Few: scan them as that kernel does.  Many (a dense region: hundreds to thousands): take the fine site's cells instead.
    bool dense = false;
    if (!done) {
        const int H8 = g.h8 * 8, W8 = g.w8 * 8;
        const int row = lane >> 1, part = lane & 1;
        const int wi = i - 2 + row;
        const int cbeg = max(j - 2, 0), cend = min(j + 2, W8 - 1);
        const int split = min(cend, (cbeg | 7));                  // last column of the first block
        const int c0 = part == 0 ? cbeg : split + 1;
        const int c1 = part == 0 ? split : cend;
        int ps = 0, pe = 0;
        if (lane < 10 && wi >= 0 && wi <= H8 - 1 && c0 <= c1) {
            ps = cellstart_c[cell_key(wi, c0, g)];
            pe = cellstart_c[cell_key(wi, c1, g) + 1];
        }
        const int total = wave_sum_i(pe - ps);
        dense = total > dense_min;
        if (!dense) {
            unsigned long long live = __ballot(pe > ps);
            while (live) {
                const int l = __ffsll((long long)live) - 1;
                live &= live - 1;
                scan_range(__builtin_amdgcn_readlane(ps, l), __builtin_amdgcn_readlane(pe, l));
            }
            wave_merge<K>(top, gd, gi, gcnt);
            const float cw = fminf(s / g.xs, s / g.ys);
            const float bound = 2.5f * cw - 1e-3f;
            if (gcnt >= K && gd[K - 1] < bound * bound) done = true;
            if (rmax2 >= 0.0f && bound > 0.f && bound * bound > rmax2) done = true;
        }
    }
    // ---- phase A (dense regions): windows of the FINE cells, radius 2, 4, 6, 8: window r adds the cells with max(|a|, |b|) in (r - 2, r]
    int rprev = -1;
    for (int r = 2; r <= 8 && !done && dense; r += 2) {
        const int side = 2 * r + 1, ncell = side * side;
        for (int t0 = 0; t0 < ncell; t0 += 64) {
            const int t = t0 + lane;
            const int a = t / side - r, b = t - (t / side) * side - r;
            const int fi = ci + a, fj = cj + b;
            if (t < ncell && max(abs(a), abs(b)) > rprev && fi >= 0 && fi < gf.h && fj >= 0 && fj < gf.w) {
                const int key = cell_key(fi, fj, gf);
                scan_own(cellstart[key], cellstart[key + 1]);
            }
        }
        rprev = r;
        wave_merge<K>(top, gd, gi, gcnt);
        const float bound = open_edge_distance(gf, X, Y, ci - r, ci + r, cj - r, cj + r);
        if (gcnt >= K && bound > 0.f && gd[K - 1] < bound * bound) done = true;
        if (rmax2 >= 0.0f && bound > 0.f && bound * bound > rmax2) done = true;
    }
    // ---- phase B: block rings of the COARSE site's own cells, as in k_knn_search_wave (fresh start: the rings re-cover the windows)
    if (!done) {
        top.clear();
        gcnt = 0;
#pragma unroll
        for (int q = 0; q < K; ++q) { gd[q] = 3.0e38f; gi[q] = 0x7fffffff; }
        const float cwmin = fminf(s / g.xs, s / g.ys);
        const int I = i >> 3, J = j >> 3;
        const int Rmax = max(g.h8, g.w8);
        for (int R = 0; R <= Rmax && !done; ++R) {
            const int nblk = R == 0 ? 1 : 8 * R;
            for (int t0 = 0; t0 < nblk; t0 += 64) {
                const int t = t0 + lane;
                int a = 0, b = 0;
                if (R > 0) {
                    if (t < 2 * R + 1) { a = -R; b = -R + t; }
                    else if (t < 4 * R + 2) { a = R; b = -R + (t - (2 * R + 1)); }
                    else if (t < 6 * R + 1) { b = -R; a = -R + 1 + (t - (4 * R + 2)); }
                    else { b = R; a = -R + 1 + (t - (6 * R + 1)); }
                }
                const int bi = I + a, bj = J + b;
                int ps = 0, pe = 0;
                if (t < nblk && bi >= 0 && bi < g.h8 && bj >= 0 && bj < g.w8) {
                    const int k0 = (bi * g.w8 + bj) << 6;
                    ps = cellstart_c[k0];
                    pe = cellstart_c[k0 + 64];
                }
                unsigned long long live = __ballot(pe > ps);
                while (live) {
                    const int l = __ffsll((long long)live) - 1;
                    live &= live - 1;
                    const int cbi = __builtin_amdgcn_readlane(bi, l), cbj = __builtin_amdgcn_readlane(bj, l);
                    const int cps = __builtin_amdgcn_readlane(ps, l), cpe = __builtin_amdgcn_readlane(pe, l);
                    bool visit = true;
                    if (gcnt >= K) {   // prune with the (possibly stale, hence conservative) wave-wide K-th
                        const float lox = ((float)(cbi * 8) * s - g.xo) / g.xs, hix = ((float)(cbi * 8 + 8) * s - g.xo) / g.xs;
                        const float loy = ((float)(cbj * 8) * s - g.yo) / g.ys, hiy = ((float)(cbj * 8 + 8) * s - g.yo) / g.ys;
                        const float ddx = fmaxf(0.f, fmaxf(lox - X, X - hix) - 1e-3f);
                        const float ddy = fmaxf(0.f, fmaxf(loy - Y, Y - hiy) - 1e-3f);
                        visit = (ddx * ddx + ddy * ddy) <= gd[K - 1];
                    }
                    if (visit) scan_range(cps, cpe);
                }
            }
            wave_merge<K>(top, gd, gi, gcnt);
            const float bound = (8.0f * (float)R + 0.5f) * cwmin - 1e-3f;
            if (gcnt >= K && gd[K - 1] < bound * bound) done = true;
            if (rmax2 >= 0.0f && bound > 0.f && bound * bound > rmax2) done = true;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) out[q * hw + pix] = (q < gcnt) ? gi[q] : -1;
    }
}

template <int K>
__global__ void __launch_bounds__(256) k_knn_search_fine(const int *count, int n_max, KnnGrid g, KnnGrid gf, const int *cellstart_c,
                                                         const float4 *sorted_c, const int *cellstart, const float4 *sorted, float rmax2,
                                                         int *out, int dense_min)
{
    knn_search_fine<K>(count, n_max, g, gf, cellstart_c, sorted_c, cellstart, sorted, rmax2, out, dense_min, blockIdx.x);
}

// The searches of ALL sites of a dcf_knn_bev_sites call in one launch (round 4): grid.x = the sites' workgroups one after the
// other, grid.y = frame.  Each site runs the kernel body it would run on its own -- tile kernel with one wave (four tiles per
// workgroup) or four waves per tile, wave kernel, fine-cell kernel -- so the maps are the same bits; what changes is that the
// four launches (66 + 66 + 39 + 39 us at cfg2, each waiting for its slowest tile with most of the chip idle) overlap.
struct KnnSearchSites {
    KnnGrid g[DCF_MAX_KNN_SITES], gf[DCF_MAX_KNN_SITES];
    const int *cellstart[DCF_MAX_KNN_SITES], *cellstart_f[DCF_MAX_KNN_SITES];
    const float4 *sorted[DCF_MAX_KNN_SITES], *sorted_f[DCF_MAX_KNN_SITES];
    int *out[DCF_MAX_KNN_SITES];
    int kind[DCF_MAX_KNN_SITES];          // 0 tile kernel, one wave per tile; 1 tile kernel, four waves per tile; 2 wave kernel; 3 fine-cell kernel
    int first[DCF_MAX_KNN_SITES + 1];     // first workgroup of each site
    int n, dense_min;
};

template <int K>
__global__ void __launch_bounds__(256) k_knn_search_ms(const int *count, int n_max, KnnSearchSites S, float rmax2)
{
    int i = 0;
#pragma unroll
    for (int k = 1; k < DCF_MAX_KNN_SITES; ++k) i += (k < S.n && (int)blockIdx.x >= S.first[k]);
    const int lb = (int)blockIdx.x - S.first[i];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const KnnGrid &g = S.g[i];
    const int kind = S.kind[i];
    if (kind == 0) {
        const int tile = lb * 4 + wave;
        if (tile >= g.h8 * g.w8) return;
        knn_search_tile<K, 1>(count, n_max, g, S.cellstart[i], S.sorted[i], rmax2, S.out[i], tile, 0);
    } else if (kind == 1) {
        knn_search_tile<K, 4>(count, n_max, g, S.cellstart[i], S.sorted[i], rmax2, S.out[i], lb, wave);
    } else if (kind == 2) {
        knn_search_wave<K>(count, n_max, g, S.cellstart[i], S.sorted[i], rmax2, S.out[i], lb);
    } else {
        knn_search_fine<K>(count, n_max, g, S.gf[i], S.cellstart[i], S.sorted[i], S.cellstart_f[i], S.sorted_f[i], rmax2, S.out[i], S.dense_min, lb);
    }
}

// ------------------------------------------------------------------ inverse of the KNN maps (for the fusion backward)
// idx [K][h][w] says which points each BEV pixel gathers; the backward wants, per point, the pixels that gathered it.
// All maps of a step (sites x frames) are inverted together: counting sort by key (map, point id):
// hist -> exclusive scan -> fill (atomic cursor).  ent_pix packs the pixel as (i << 16) | j.
struct InvMaps {
    const int *idx[DCF_MAX_KNN_MAPS];
    int first[DCF_MAX_KNN_MAPS + 1];   // first pair slot of each map in the concatenated (K*h*w) space
    int hw[DCF_MAX_KNN_MAPS], w[DCF_MAX_KNN_MAPS];
    int n;
};

__device__ __forceinline__ int inv_map_of(const InvMaps &m, int e)
{
    int g = 0;
#pragma unroll
    for (int i = 1; i < DCF_MAX_KNN_MAPS; ++i) g += (i < m.n && e >= m.first[i]);
    return g;
}

// Neighbouring pixels mostly share their nearest points (a far, isolated point owns thousands of pixels), so the
// lanes of a wave -- consecutive pixels of one row -- are grouped into runs of equal (map, id): one atomic per run,
// issued by its first lane, instead of one same-address atomic per pixel.
struct InvRun { int head, len; };
__device__ __forceinline__ InvRun inv_run(int key, int lane)
{
    const int prev = __shfl_up(key, 1, 64);
    const unsigned long long heads = __ballot(lane == 0 || key != prev);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);          // heads at or below this lane
    InvRun r;
    r.head = 63 - __clzll(below);
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
    r.len = (above ? lane + 1 + (__ffsll((long long)above) - 1) : 64) - r.head;    // valid on the head lane
    return r;
}

// zero `n` ints in each of gridDim.y rows `stride` ints apart (hipMemset2DAsync took 14 us for 2 x 560 KB)
__global__ void __launch_bounds__(256) k_zero_rows(int *p, int n, int64_t stride)
{
    int4 *row = reinterpret_cast<int4 *>(p + blockIdx.y * stride);
    const int n4 = n >> 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) row[i] = make_int4(0, 0, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[blockIdx.y * stride + (n4 << 2) + threadIdx.x] = 0;
}

// (map, index inside the map) of pair slot e.  A block of 256 slots almost never straddles two maps (a map has K*h*w slots): the
// map of the block's first and last slot are found with block-uniform (scalar) arithmetic, and only a straddling block pays the
// 31 compares per thread -- they were a third of these kernels' instructions (round 5).
__device__ __forceinline__ int inv_load(const InvMaps &m, int e, int &g)
{
    const int e0 = blockIdx.x * blockDim.x;
    const int gA = __builtin_amdgcn_readfirstlane(inv_map_of(m, e0));
    const int gB = __builtin_amdgcn_readfirstlane(inv_map_of(m, min(e0 + (int)blockDim.x, m.first[m.n]) - 1));
    if (gA == gB) {
        g = gA;
        return m.idx[gA][e - m.first[gA]];
    }
    g = inv_map_of(m, e);
    return m.idx[g][e - m.first[g]];
}

__global__ void __launch_bounds__(256) k_inv_hist(InvMaps m, int n_max, int *cnt)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int key = -1;
    if (e < m.first[m.n]) {
        int g;
        const int id = inv_load(m, e, g);
        if (id >= 0 && id < n_max) key = g * (n_max + 1) + id;
    }
    const InvRun r = inv_run(key, lane);
    if (key >= 0 && r.head == lane) atomicAdd(&cnt[key], r.len);
}

__global__ void __launch_bounds__(256) k_inv_fill(InvMaps m, int n_max, int *cursor, int *ent_pix, int *ent_pt)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int key = -1, g = 0, id = 0;
    if (e < m.first[m.n]) {
        id = inv_load(m, e, g);
        if (id >= 0 && id < n_max) key = g * (n_max + 1) + id;
    }
    const InvRun r = inv_run(key, lane);
    int base = 0;
    if (key >= 0 && r.head == lane) base = atomicAdd(&cursor[key], r.len);
    base = __shfl(base, r.head, 64);
    if (key < 0) return;
    const int pos = base + (lane - r.head);
    const int p = (e - m.first[g]) % m.hw[g];
    ent_pix[pos] = ((p / m.w[g]) << 16) | (p % m.w[g]);
    ent_pt[pos] = id;
}

}  // namespace

// ================================================================== C ABI
extern "C" size_t dcf_compact_workspace_bytes(int n) { return sizeof(int) * (size_t)(cdiv(n > 0 ? n : 1, CP_TILE) + 8); }

extern "C" int dcf_range_filter(const float *pts, int n, const float *lim, float *out_pts, int32_t *out_src,
                                int32_t *count_dev, void *ws, dcf_stream_t stream)
{
    DCF_REQUIRE(n >= 0 && lim && count_dev && ws, "dcf_range_filter: bad arguments");
    hipStream_t s = S(stream);
    if (n == 0) { DCF_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), s)); return DCF_OK; }
    RangePred p;
    memcpy(p.lim.v, lim, sizeof(p.lim.v));
    int nb = cdiv(n, CP_TILE);
    int *bs = (int *)ws;
    DCF_LAUNCH_B("compact_count", (double)n * 12.0, s, hipLaunchKernelGGL(k_compact_count<RangePred>, dim3(nb), dim3(CP_THREADS), 0, s, pts, n, p, bs));
    DCF_LAUNCH("compact_scan", s, hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(CP_THREADS), 0, s, bs, nb, count_dev));
    DCF_LAUNCH_B("compact_scatter", (double)n * (12.0 + 16.0), s, hipLaunchKernelGGL((k_compact_scatter<RangePred, false>), dim3(nb), dim3(CP_THREADS), 0, s, pts, n, p, bs,
                                                         (float *)nullptr, out_pts, out_src));
    return DCF_OK;
}

extern "C" int dcf_project_filter(const float *pts, int n, const float *lim, const float *crt, float ulim, float vlim,
                                  int mode, float *uv_out, float *xyz_out, int32_t *src_out, int32_t *count_dev,
                                  void *ws, dcf_stream_t stream)
{
    DCF_REQUIRE(n >= 0 && lim && crt && count_dev && ws && uv_out && xyz_out, "dcf_project_filter: bad arguments");
    hipStream_t s = S(stream);
    if (n == 0) { DCF_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), s)); return DCF_OK; }
    ProjPred p;
    memcpy(p.lim.v, lim, sizeof(p.lim.v));
    memcpy(p.c.v, crt, sizeof(p.c.v));
    p.ulim = ulim; p.vlim = vlim; p.mode = mode;
    int nb = cdiv(n, CP_TILE);
    int *bs = (int *)ws;
    DCF_LAUNCH_B("project_count", (double)n * 12.0, s, hipLaunchKernelGGL(k_compact_count<ProjPred>, dim3(nb), dim3(CP_THREADS), 0, s, pts, n, p, bs));
    DCF_LAUNCH("compact_scan", s, hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(CP_THREADS), 0, s, bs, nb, count_dev));
    DCF_LAUNCH_B("project_scatter", (double)n * (12.0 + 20.0), s, hipLaunchKernelGGL((k_compact_scatter<ProjPred, true>), dim3(nb), dim3(CP_THREADS), 0, s, pts, n, p, bs,
                                                         uv_out, xyz_out, src_out));
    return DCF_OK;
}

// The same for the B frames of a batch in one launch per phase.  pts / n / crt: HOST arrays (B device pointers, B counts, B x 12
// floats); uv_out [B][rows][2], xyz_out [B][rows][3] (rows >= every n[b]; rows past a frame's count are left untouched), count_dev
// [B]; ws: B * dcf_compact_workspace_bytes(max n).  Same results as B dcf_project_filter calls, bit for bit.
extern "C" int dcf_project_filter_batch(const float *const *pts, const int *n, int B, const float *lim, const float *crt, float ulim, float vlim,
                                        int mode, float *uv_out, float *xyz_out, int rows, int32_t *count_dev, void *ws, dcf_stream_t stream)
{
    const char *who = "dcf_project_filter_batch";
    DCF_REQUIRE(pts && n && lim && crt && count_dev && ws && uv_out && xyz_out, "%s: bad arguments", who);
    DCF_REQUIRE(B >= 1 && B <= DCF_PROJ_BATCH_MAX, "%s: 1..%d frames per call", who, DCF_PROJ_BATCH_MAX);
    hipStream_t s = S(stream);
    ProjBatch pb;
    int nmax = 0;
    for (int b = 0; b < DCF_PROJ_BATCH_MAX; ++b) {
        pb.pts[b] = b < B ? pts[b] : nullptr;
        pb.n[b] = b < B ? n[b] : 0;
        DCF_REQUIRE(b >= B || (n[b] >= 0 && n[b] <= rows && (n[b] == 0 || pts[b])), "%s: frame %d: bad point count / null points", who, b);
        memcpy(pb.c[b].v, crt + 12 * (b < B ? b : 0), sizeof(pb.c[b].v));
        if (b < B && n[b] > nmax) nmax = n[b];
    }
    if (nmax == 0) { DCF_HIP(hipMemsetAsync(count_dev, 0, sizeof(int) * B, s)); return DCF_OK; }
    Lim6 l;
    memcpy(l.v, lim, sizeof(l.v));
    const int nb = cdiv(nmax, CP_TILE);
    const int bstride = (int)(dcf_compact_workspace_bytes(nmax) / sizeof(int));
    int *bs = (int *)ws;
    DCF_LAUNCH_B("project_count", (double)nmax * B * 12.0, s, hipLaunchKernelGGL(k_proj_count_b, dim3(nb, B), dim3(CP_THREADS), 0, s, pb, l, ulim, vlim, mode, bs, bstride));
    DCF_LAUNCH("compact_scan", s, hipLaunchKernelGGL(k_proj_scan_b, dim3(1, B), dim3(CP_THREADS), 0, s, bs, nb, bstride, count_dev));
    DCF_LAUNCH_B("project_scatter", (double)nmax * B * (12.0 + 20.0), s, hipLaunchKernelGGL(k_proj_scatter_b, dim3(nb, B), dim3(CP_THREADS), 0, s, pb, l, ulim, vlim, mode, bs, bstride,
                                                                                    uv_out, xyz_out, rows));
    return DCF_OK;
}

extern "C" size_t dcf_voxelize_workspace_bytes(int Cz, int L, int W) { return sizeof(int) * 2 * (size_t)Cz * L * W; }


extern "C" int dcf_voxelize(const float *pts, int n, const float *lim, const float *aff, int Cz, int L, int W,
                            int mode, float *grid, void *owner_ws, dcf_stream_t stream)
{
    DCF_REQUIRE(n >= 0 && lim && aff && grid && Cz > 0 && L > 0 && W > 0, "dcf_voxelize: bad arguments");
    DCF_REQUIRE((int64_t)Cz * L * W < (1ll << 31), "dcf_voxelize: grid too large for int32 voxel ids");
    hipStream_t s = S(stream);
    const int nvox = Cz * L * W;
    DCF_HIP(hipMemsetAsync(grid, 0, sizeof(float) * (size_t)nvox, s));
    if (n == 0) return DCF_OK;
    Lim6 l; Aff6 a;
    memcpy(l.v, lim, sizeof(l.v));
    memcpy(a.v, aff, sizeof(a.v));
    const int nb = cdiv(n, 256);
    if (mode == DCF_VOXEL_COMPAT) {
        DCF_REQUIRE(owner_ws != nullptr, "dcf_voxelize: compat mode needs the zeroed owner workspace");
        VoxBatch vb;
        for (int b = 0; b < DCF_MAX_VOXEL_BATCH; ++b) { vb.pts[b] = b == 0 ? pts : nullptr; vb.n[b] = b == 0 ? n : 0; }
        const dim3 g1(nb, 1);
        DCF_LAUNCH("voxel_cell_claim", s, hipLaunchKernelGGL(k_voxel_cell_claim, g1, dim3(256), 0, s, vb, l, a, L, W, nvox, (int *)owner_ws, 0));
        DCF_LAUNCH("voxel_cell_gather", s, hipLaunchKernelGGL((k_voxel_cell_gather<float, false>), g1, dim3(256), 0, s, vb, l, a, Cz, L, W, nvox, grid, (const int *)owner_ws));
        DCF_LAUNCH("voxel_cell_claim", s, hipLaunchKernelGGL(k_voxel_cell_claim, g1, dim3(256), 0, s, vb, l, a, L, W, nvox, (int *)owner_ws, 1));
    } else if (mode == DCF_VOXEL_COMPAT_ROUNDS) {
        DCF_REQUIRE(owner_ws != nullptr, "dcf_voxelize: compat mode needs the zeroed owner workspace");
        for (int r = 0; r <= 8; ++r)
            DCF_LAUNCH("voxel_compat_round", s, hipLaunchKernelGGL(k_voxel_compat_round, dim3(nb), dim3(256), 0, s, pts, n, l, a, L, W, nvox, r,
                                                                   grid, (int *)owner_ws));
    } else if (mode == DCF_VOXEL_ACCUM) {
        DCF_LAUNCH("voxel_accum", s, hipLaunchKernelGGL(k_voxel_accum, dim3(nb), dim3(256), 0, s, pts, n, l, a, L, W, grid));
    } else if (mode == DCF_VOXEL_OCCUPANCY) {
        DCF_LAUNCH("voxel_occupancy", s, hipLaunchKernelGGL(k_voxel_occupancy, dim3(nb), dim3(256), 0, s, pts, n, l, a, L, W, grid));
    } else {
        dcf_set_error("dcf_voxelize: unknown mode %d", mode);
        return DCF_EINVAL;
    }
    return DCF_OK;
}

static int voxelize_batch_impl(const char *who, int dtype, bool nhwc, const float *const *pts, const int *n, int B, const float *lim, const float *aff,
                               int Cz, int L, int W, void *grids, void *owner_ws, hipStream_t s)
{
    DCF_REQUIRE(pts && n && lim && aff && grids && owner_ws && Cz > 0 && L > 0 && W > 0, "%s: bad arguments", who);
    DCF_REQUIRE(B >= 1 && B <= DCF_MAX_VOXEL_BATCH, "%s: 1..%d frames per call", who, DCF_MAX_VOXEL_BATCH);
    DCF_REQUIRE((int64_t)Cz * L * W < (1ll << 31), "%s: grid too large for int32 voxel ids", who);
    const int nvox = Cz * L * W;
    const size_t es = dtype == DCF_F32 ? 4 : 2;
    DCF_HIP(hipMemsetAsync(grids, 0, es * (size_t)nvox * B, s));
    VoxBatch vb;
    int nmax = 0;
    for (int b = 0; b < DCF_MAX_VOXEL_BATCH; ++b) {
        vb.pts[b] = b < B ? pts[b] : nullptr;
        vb.n[b] = b < B ? n[b] : 0;
        DCF_REQUIRE(b >= B || (n[b] >= 0 && (n[b] == 0 || pts[b])), "%s: frame %d: null points", who, b);
        if (b < B && n[b] > nmax) nmax = n[b];
    }
    if (nmax == 0) return DCF_OK;
    Lim6 l; Aff6 a;
    memcpy(l.v, lim, sizeof(l.v));
    memcpy(a.v, aff, sizeof(a.v));
    const dim3 grid(cdiv(nmax, 256), B);
    DCF_LAUNCH_B("voxel_cell_claim", (double)nmax * B * 16.0, s, hipLaunchKernelGGL(k_voxel_cell_claim, grid, dim3(256), 0, s, vb, l, a, L, W, nvox, (int *)owner_ws, 0));
    if (!nhwc) {
        DCF_LAUNCH_B("voxel_cell_gather", (double)nmax * B * (12.0 + 27 * 4.0 + 8 * 4.0), s, hipLaunchKernelGGL((k_voxel_cell_gather<float, false>), grid, dim3(256), 0, s, vb, l, a, Cz, L, W, nvox, (float *)grids, (const int *)owner_ws));
    } else {
        DCF_DISPATCH_DTYPE(dtype, { DCF_LAUNCH_B("voxel_cell_gather", (double)nmax * B * (12.0 + 27 * 4.0 + 8.0 * sizeof(T)), s, hipLaunchKernelGGL((k_voxel_cell_gather<T, true>), grid, dim3(256), 0, s, vb, l, a, Cz, L, W, nvox, (T *)grids, (const int *)owner_ws)); })
    }
    DCF_LAUNCH_B("voxel_cell_claim", (double)nmax * B * 16.0, s, hipLaunchKernelGGL(k_voxel_cell_claim, grid, dim3(256), 0, s, vb, l, a, L, W, nvox, (int *)owner_ws, 1));
    return DCF_OK;
}

extern "C" int dcf_voxelize_batch(const float *const *pts, const int *n, int B, const float *lim, const float *aff, int Cz, int L, int W,
                                  float *grids, void *owner_ws, dcf_stream_t stream)
{
    return voxelize_batch_impl("dcf_voxelize_batch", DCF_F32, false, pts, n, B, lim, aff, Cz, L, W, grids, owner_ws, S(stream));
}

extern "C" int dcf_voxelize_batch_nhwc(int dtype, const float *const *pts, const int *n, int B, const float *lim, const float *aff, int Cz, int L,
                                       int W, void *x_nhwc, void *owner_ws, dcf_stream_t stream)
{
    DCF_REQUIRE(dtype == DCF_F32 || dtype == DCF_BF16 || dtype == DCF_F16, "dcf_voxelize_batch_nhwc: unsupported dtype %d", dtype);
    return voxelize_batch_impl("dcf_voxelize_batch_nhwc", dtype, true, pts, n, B, lim, aff, Cz, L, W, x_nhwc, owner_ws, S(stream));
}

static inline void knn_dims(int h, int w, int &h8, int &w8, int &ncell) { h8 = (h + 7) / 8; w8 = (w + 7) / 8; ncell = h8 * w8 * 64; }

// workspace layout (ints): cellcnt[ncell+1] | cellstart[ncell+1] | cursor[ncell+1] | blocksum[nb+8] | pkey[n_max] | sorted float4[n_max]
extern "C" size_t dcf_knn_workspace_bytes(int n_max, int h, int w)
{
    int h8, w8, ncell;
    knn_dims(h, w, h8, w8, ncell);
    size_t ints = 3 * (size_t)(ncell + 1) + (size_t)cdiv(ncell + 1, CP_TILE) + 8 + (size_t)n_max;
    ints = (ints + 3) & ~(size_t)3;
    return ints * sizeof(int) + sizeof(float4) * (size_t)n_max + 64;
}

struct KnnFine {                 // a finer site of the same batch whose cells are already built (dcf_knn_bev_batch_shared)
    int h, w, stride;
    const void *ws;
    size_t ws_stride_bytes;
};

// What the search of one site runs: which kernel, with what of the finer site (dcf_knn_bev_sites defers the launch and issues
// all sites' searches as one: k_knn_search_ms).
struct KnnSearchPlan {
    KnnGrid g, gf;
    const int *cellstart, *cellstart_f;
    const float4 *sorted, *sorted_f;
    int kind;           // 0 / 1: tile kernel (tile_waves 1 or 2 / 4), 2: wave kernel, 3: fine-cell kernel
    int tile_waves, dense_min;
};

static KnnSearchPlan knn_search_plan(const KnnGrid &g, int B, int n_max, int K, const int *cellstart, const float4 *sorted, const KnnFine *fine, bool merged)
{
    KnnSearchPlan pl;
    pl.g = g; pl.gf = g; pl.cellstart = cellstart; pl.sorted = sorted; pl.cellstart_f = nullptr; pl.sorted_f = nullptr;
    pl.tile_waves = 1; pl.dense_min = KNN_FINE_MIN_POINTS;
    // coarse sites: one wave per pixel (lanes split the candidates).  DCF_KNN_KERNEL = wave | tile forces one of the two
    // kernels (dcf_set_option: the parity tests compare them on the same site; both produce the exact (d2, index) order)
    static DcfOpt force_o("KNN_KERNEL"); const char *force = force_o.str();
    const bool per_wave = force && force[0] == 'w' ? true : (force && force[0] == 't' ? false : (g.h * g.w <= 20000));
    // waves per tile of k_knn_search (option KNN_TILE_WAVES = 1 | 2 | 4 forces)
    static DcfOpt tw_o("KNN_TILE_WAVES"); const char *tw = tw_o.str();
    // automatic: four waves for a launch of up to 1500 tiles on its own; one in the all-sites launch, which fills the chip anyway
    // (121 against 127 us at cfg2, 184 with four everywhere)
    pl.tile_waves = tw && (tw[0] == '1' || tw[0] == '2' || tw[0] == '4') ? tw[0] - '0' : (!merged && g.h8 * g.w8 * B <= 1500 ? 4 : 1);
    if (fine) {
        // dense pixels on the fine site's cells, the rest on this site's own blocks (k_knn_search_fine)
        KnnGrid gf = g;
        gf.h = fine->h; gf.w = fine->w; gf.stride = fine->stride;
        int ncf;
        knn_dims(fine->h, fine->w, gf.h8, gf.w8, ncf);
        gf.fs_ws = B > 1 ? (int)(fine->ws_stride_bytes / 4) : 0;
        const int nscf = ncf + 1, nsbf = cdiv(nscf, CP_TILE);
        size_t intsf = 3 * (size_t)nscf + (size_t)nsbf + 8 + (size_t)n_max;
        intsf = (intsf + 3) & ~(size_t)3;
        static DcfOpt dm_o("KNN_FINE_MIN"); const char *dm = dm_o.str();
        pl.gf = gf;
        pl.cellstart_f = (const int *)fine->ws + nscf;
        pl.sorted_f = (const float4 *)((const char *)fine->ws + intsf * sizeof(int));
        pl.dense_min = dm ? atoi(dm) : KNN_FINE_MIN_POINTS;
        pl.kind = 3;
    } else if (per_wave) pl.kind = 2;
    else pl.kind = pl.tile_waves == 4 ? 1 : 0;
    (void)K;
    return pl;
}

static int knn_bev_impl(const char *who, const float *xyz, const int32_t *count_dev, int B, int n_max, int K, int h, int w, int stride,
                        float xs, float xo, float ys, float yo, float rmax2, int32_t *idx_out, void *ws, size_t ws_stride_bytes, hipStream_t s,
                        const KnnFine *fine = nullptr, bool presorted = false, KnnSearchPlan *defer = nullptr)
{
    DCF_REQUIRE(xyz && count_dev && idx_out && ws, "%s: null pointer", who);
    DCF_REQUIRE(K >= 1 && K <= 8, "%s: K must be 1..8 (got %d)", who, K);
    DCF_REQUIRE(h > 0 && w > 0 && stride > 0 && n_max >= 0 && B >= 1 && B <= 65535, "%s: bad dims", who);
    DCF_REQUIRE(B == 1 || (ws_stride_bytes % 16 == 0 && ws_stride_bytes >= dcf_knn_workspace_bytes(n_max, h, w) && ws_stride_bytes / 4 < (1ull << 31)),
                "%s: workspace stride must be a multiple of 16 bytes and hold one frame's workspace", who);
    KnnGrid g;
    g.h = h; g.w = w; g.stride = stride; g.xs = xs; g.xo = xo; g.ys = ys; g.yo = yo;
    g.fs_xyz = B > 1 ? n_max * 3 : 0; g.fs_cnt = B > 1 ? 1 : 0; g.fs_ws = B > 1 ? (int)(ws_stride_bytes / 4) : 0; g.fs_out = B > 1 ? K * h * w : 0;
    int ncell;
    knn_dims(h, w, g.h8, g.w8, ncell);
    const int nscan = ncell + 1;
    const int nsb = cdiv(nscan, CP_TILE);
    int *cellcnt = (int *)ws;
    int *cellstart = cellcnt + nscan;
    int *cursor = cellstart + nscan;
    int *blocksum = cursor + nscan;
    int *pkey = blocksum + nsb + 8;
    size_t ints = 3 * (size_t)nscan + (size_t)nsb + 8 + (size_t)n_max;
    ints = (ints + 3) & ~(size_t)3;
    float4 *sorted = (float4 *)((char *)ws + ints * sizeof(int));
    const double fB = (double)B;
    if (!presorted) {                    // (dcf_knn_bev_sites has built the tables of every site already)
        if (B == 1) DCF_HIP(hipMemsetAsync(cellcnt, 0, sizeof(int) * (size_t)nscan, s));
        else if ((ws_stride_bytes & 15) == 0 && ((uintptr_t)cellcnt & 15) == 0)
            DCF_LAUNCH_B("knn_zero", (double)B * nscan * 4.0, s, hipLaunchKernelGGL(k_zero_rows, dim3(std::min(cdiv(nscan, 1024), 256), B), dim3(256), 0, s, cellcnt, nscan, (int64_t)(ws_stride_bytes / 4)));
        else DCF_HIP(hipMemset2DAsync(cellcnt, ws_stride_bytes, 0, sizeof(int) * (size_t)nscan, B, s));
        if (n_max > 0) {
            const int nb = cdiv(n_max, 256);
            DCF_LAUNCH_B("knn_hist", fB * n_max * 16.0, s, hipLaunchKernelGGL(k_knn_hist, dim3(nb, B), dim3(256), 0, s, xyz, count_dev, n_max, g, cellcnt, pkey));
        }
        int *total = blocksum + nsb;  // scratch int for the scan total
        DCF_LAUNCH("scan_blocksum", s, hipLaunchKernelGGL(k_scan_blocksum, dim3(nsb, B), dim3(CP_THREADS), 0, s, cellcnt, nscan, blocksum, g.fs_ws));
        DCF_LAUNCH("compact_scan", s, hipLaunchKernelGGL(k_compact_scan, dim3(1, B), dim3(CP_THREADS), 0, s, blocksum, nsb, total, g.fs_ws));
        DCF_LAUNCH("scan_apply", s, hipLaunchKernelGGL(k_scan_apply, dim3(nsb, B), dim3(CP_THREADS), 0, s, cellcnt, nscan, blocksum, cellstart, cursor, g.fs_ws));
        if (n_max > 0) {
            const int nb = cdiv(n_max, 256);
            DCF_LAUNCH_B("knn_fill", fB * n_max * (12.0 + 4.0 + 16.0), s, hipLaunchKernelGGL(k_knn_fill, dim3(nb, B), dim3(256), 0, s, xyz, count_dev, n_max, pkey, cursor, sorted, g.fs_xyz, g.fs_cnt, g.fs_ws));
        }
    }
    const KnnSearchPlan pl = knn_search_plan(g, B, n_max, K, cellstart, sorted, fine, defer != nullptr);
    if (defer) { *defer = pl; return DCF_OK; }
    const int nbp = g.h8 * g.w8;            // one workgroup (one, two or four waves) per 8x8 pixel tile
    const int nbw = cdiv(h * w, 4);         // one wave per pixel
    if (pl.kind == 3) {
#define KNN_FCASE(KK)                                                                                                                       \
    case KK:                                                                                                                                \
        DCF_LAUNCH_B("knn_search_fine", fB * ((double)h * w * K * 4.0 + (double)n_max * 16.0), s,                                           \
                     hipLaunchKernelGGL(k_knn_search_fine<KK>, dim3(nbw, B), dim3(256), 0, s, count_dev, n_max, g, pl.gf, cellstart, sorted,   \
                                        pl.cellstart_f, pl.sorted_f, rmax2, idx_out, pl.dense_min));                                        \
        break;
        switch (K) {
            KNN_FCASE(1) KNN_FCASE(2) KNN_FCASE(3) KNN_FCASE(4) KNN_FCASE(5) KNN_FCASE(6) KNN_FCASE(7) KNN_FCASE(8)
        }
#undef KNN_FCASE
        return DCF_OK;
    }
#define KNN_CASE(KK)                                                                                                     \
    case KK:                                                                                                             \
        if (pl.kind == 2)                                                                                                \
            DCF_LAUNCH_B("knn_search_wave", fB * ((double)h * w * K * 4.0 + (double)n_max * 16.0), s, hipLaunchKernelGGL(k_knn_search_wave<KK>, dim3(nbw, B), dim3(256), 0, s, count_dev, n_max, g, \
                                                                cellstart, sorted, rmax2, idx_out));                     \
        else if (pl.tile_waves == 4)                                                                                     \
            DCF_LAUNCH_B("knn_search", fB * ((double)h * w * K * 4.0 + (double)n_max * 16.0), s, hipLaunchKernelGGL((k_knn_search<KK, 4>), dim3(nbp, B), dim3(256), 0, s, count_dev, n_max, g, \
                                                           cellstart, sorted, rmax2, idx_out));                          \
        else if (pl.tile_waves == 2)                                                                                     \
            DCF_LAUNCH_B("knn_search", fB * ((double)h * w * K * 4.0 + (double)n_max * 16.0), s, hipLaunchKernelGGL((k_knn_search<KK, 2>), dim3(nbp, B), dim3(128), 0, s, count_dev, n_max, g, \
                                                           cellstart, sorted, rmax2, idx_out));                          \
        else                                                                                                             \
            DCF_LAUNCH_B("knn_search", fB * ((double)h * w * K * 4.0 + (double)n_max * 16.0), s, hipLaunchKernelGGL((k_knn_search<KK, 1>), dim3(nbp, B), dim3(64), 0, s, count_dev, n_max, g, \
                                                           cellstart, sorted, rmax2, idx_out));                          \
        break;
    switch (K) {
        KNN_CASE(1) KNN_CASE(2) KNN_CASE(3) KNN_CASE(4) KNN_CASE(5) KNN_CASE(6) KNN_CASE(7) KNN_CASE(8)
    }
#undef KNN_CASE
    return DCF_OK;
}

extern "C" int dcf_knn_bev(const float *xyz, const int32_t *count_dev, int n_max, int K, int h, int w, int stride,
                           float xs, float xo, float ys, float yo, float rmax2, int32_t *idx_out, void *ws,
                           dcf_stream_t stream)
{
    return knn_bev_impl("dcf_knn_bev", xyz, count_dev, 1, n_max, K, h, w, stride, xs, xo, ys, yo, rmax2, idx_out, ws, 0, S(stream));
}

// B frames of a batch in one launch per phase (grid.y = frame): xyz [B][n_max][3], count_dev [B], idx_out [B][K][h][w], ws = B
// workspaces ws_stride_bytes apart.  Same results as B calls of dcf_knn_bev; a coarse site's search then fills the chip.
extern "C" int dcf_knn_bev_batch(const float *xyz, const int32_t *count_dev, int B, int n_max, int K, int h, int w, int stride,
                                 float xs, float xo, float ys, float yo, float rmax2, int32_t *idx_out, void *ws, size_t ws_stride_bytes,
                                 dcf_stream_t stream)
{
    return knn_bev_impl("dcf_knn_bev_batch", xyz, count_dev, B, n_max, K, h, w, stride, xs, xo, ys, yo, rmax2, idx_out, ws, ws_stride_bytes, S(stream));
}

// A coarser site of the same batch: its own cell sort as dcf_knn_bev_batch, but the search (k_knn_search_fine) serves the pixels in
// dense regions from the cells a FINER site's call has already built.  ws_fine = the workspace that call filled (same xyz, count,
// n_max, B), untouched since.
extern "C" int dcf_knn_bev_batch_shared(const float *xyz, const int32_t *count_dev, int B, int n_max, int K, int h, int w, int stride,
                                        int fine_h, int fine_w, int fine_stride, float xs, float xo, float ys, float yo, float rmax2,
                                        int32_t *idx_out, void *ws, size_t ws_stride_bytes, const void *ws_fine, size_t ws_fine_stride_bytes,
                                        dcf_stream_t stream)
{
    const char *who = "dcf_knn_bev_batch_shared";
    DCF_REQUIRE(ws_fine && fine_h > 0 && fine_w > 0 && fine_stride > 0, "%s: bad fine-site arguments", who);
    DCF_REQUIRE(stride % fine_stride == 0 && h * stride <= fine_h * fine_stride && w * stride <= fine_w * fine_stride,
                "%s: the coarse site must lie on the fine site's grid", who);
    DCF_REQUIRE(B == 1 || (ws_fine_stride_bytes % 16 == 0 && ws_fine_stride_bytes >= dcf_knn_workspace_bytes(n_max, fine_h, fine_w) &&
                           ws_fine_stride_bytes / 4 < (1ull << 31)), "%s: fine workspace stride must be that of the fine site's call", who);
    const KnnFine fine = {fine_h, fine_w, fine_stride, ws_fine, ws_fine_stride_bytes};
    return knn_bev_impl(who, xyz, count_dev, B, n_max, K, h, w, stride, xs, xo, ys, yo, rmax2, idx_out, ws, ws_stride_bytes, S(stream), &fine);
}

// All fusion sites of a batch in one call: the cell sort of EVERY site in one launch per phase (zero / histogram / three scan
// phases / fill: 6 launches instead of 6 per site), then each site's search exactly as dcf_knn_bev_batch /
// dcf_knn_bev_batch_shared would run it (sites[i].fine = index of an earlier, finer site of this call whose cells serve its dense
// pixels, or -1).  Same maps, bit for bit.  sites is a HOST array.
extern "C" int dcf_knn_bev_sites(const float *xyz, const int32_t *count_dev, int B, int n_max, int K, const dcf_knn_site *sites, int nsites,
                                 float xs, float xo, float ys, float yo, float rmax2, dcf_stream_t stream)
{
    const char *who = "dcf_knn_bev_sites";
    DCF_REQUIRE(xyz && count_dev && sites && nsites >= 1 && nsites <= DCF_MAX_KNN_SITES, "%s: 1..%d sites", who, DCF_MAX_KNN_SITES);
    DCF_REQUIRE(K >= 1 && K <= 8 && n_max >= 0 && B >= 1 && B * nsites <= 65535, "%s: bad dims", who);
    hipStream_t s = S(stream);
    KnnSortSites ss;
    ss.n = nsites; ss.B = B;
    int max_nsb = 1, max_nscan = 1;
    for (int i = 0; i < DCF_MAX_KNN_SITES; ++i) {
        const dcf_knn_site &t = sites[i < nsites ? i : 0];
        DCF_REQUIRE(t.idx_out && t.ws && t.h > 0 && t.w > 0 && t.stride > 0, "%s: site %d: bad arguments", who, i);
        DCF_REQUIRE(((uintptr_t)t.ws & 15) == 0 && t.ws_stride_bytes % 16 == 0 && t.ws_stride_bytes >= dcf_knn_workspace_bytes(n_max, t.h, t.w) &&
                    t.ws_stride_bytes / 4 < (1ull << 31), "%s: site %d: workspace must be 16-byte aligned, its stride a multiple of 16 bytes that holds one frame's workspace", who, i);
        DCF_REQUIRE(t.fine < i && t.fine >= -1, "%s: site %d: `fine` must name an earlier site of the call (or -1)", who, i);
        KnnGrid &g = ss.g[i];
        g.h = t.h; g.w = t.w; g.stride = t.stride; g.xs = xs; g.xo = xo; g.ys = ys; g.yo = yo;
        g.fs_xyz = B > 1 ? n_max * 3 : 0; g.fs_cnt = B > 1 ? 1 : 0; g.fs_ws = B > 1 ? (int)(t.ws_stride_bytes / 4) : 0; g.fs_out = B > 1 ? K * t.h * t.w : 0;
        int ncell;
        knn_dims(t.h, t.w, g.h8, g.w8, ncell);
        ss.ws[i] = (int *)t.ws;
        ss.nscan[i] = ncell + 1;
        ss.nsb[i] = cdiv(ncell + 1, CP_TILE);
        ss.pkey_off[i] = 3 * ss.nscan[i] + ss.nsb[i] + 8;
        size_t ints = 3 * (size_t)ss.nscan[i] + (size_t)ss.nsb[i] + 8 + (size_t)n_max;
        ints = (ints + 3) & ~(size_t)3;
        ss.sorted_off[i] = (int)ints;
        if (i < nsites) { max_nsb = std::max(max_nsb, ss.nsb[i]); max_nscan = std::max(max_nscan, ss.nscan[i]); }
    }
    const int gy = nsites * B;
    DCF_LAUNCH_B("knn_zero", (double)B * max_nscan * 4.0, s, hipLaunchKernelGGL(k_knn_zero_ms, dim3(std::min(cdiv(max_nscan, 1024), 256), gy), dim3(256), 0, s, ss));
    if (n_max > 0)
        DCF_LAUNCH_B("knn_hist", (double)gy * n_max * 16.0, s, hipLaunchKernelGGL(k_knn_hist_ms, dim3(cdiv(n_max, 256), gy), dim3(256), 0, s, xyz, count_dev, n_max, ss));
    DCF_LAUNCH("scan_blocksum", s, hipLaunchKernelGGL(k_scan_blocksum_ms, dim3(max_nsb, gy), dim3(CP_THREADS), 0, s, ss));
    DCF_LAUNCH("compact_scan", s, hipLaunchKernelGGL(k_compact_scan_ms, dim3(1, gy), dim3(CP_THREADS), 0, s, ss));
    DCF_LAUNCH("scan_apply", s, hipLaunchKernelGGL(k_scan_apply_ms, dim3(max_nsb, gy), dim3(CP_THREADS), 0, s, ss));
    if (n_max > 0)
        DCF_LAUNCH_B("knn_fill", (double)gy * n_max * (12.0 + 4.0 + 16.0), s, hipLaunchKernelGGL(k_knn_fill_ms, dim3(cdiv(n_max, 256), gy), dim3(256), 0, s, xyz, count_dev, n_max, ss));
    // the searches: one launch for all sites (option KNN_MERGED_SEARCH=0: one launch per site, as in round 3)
    static DcfOpt ms_o("KNN_MERGED_SEARCH"); const char *ms = ms_o.str();
    const bool merged = !(ms && ms[0] == '0');
    KnnSearchSites S;
    S.n = nsites; S.dense_min = KNN_FINE_MIN_POINTS;
    int blocks = 0;
    double bytes = 0.0;
    for (int i = 0; i < nsites; ++i) {
        const dcf_knn_site &t = sites[i];
        KnnFine fine;
        if (t.fine >= 0) {
            const dcf_knn_site &f = sites[t.fine];
            DCF_REQUIRE(t.stride % f.stride == 0 && t.h * t.stride <= f.h * f.stride && t.w * t.stride <= f.w * f.stride,
                        "%s: site %d must lie on the grid of its fine site", who, i);
            fine = {f.h, f.w, f.stride, f.ws, f.ws_stride_bytes};
        }
        KnnSearchPlan pl;
        int rc = knn_bev_impl(who, xyz, count_dev, B, n_max, K, t.h, t.w, t.stride, xs, xo, ys, yo, rmax2, t.idx_out, t.ws, t.ws_stride_bytes, s,
                              t.fine >= 0 ? &fine : nullptr, true, merged ? &pl : nullptr);
        if (rc) return rc;
        if (!merged) continue;
        S.g[i] = pl.g; S.gf[i] = pl.gf; S.cellstart[i] = pl.cellstart; S.sorted[i] = pl.sorted; S.cellstart_f[i] = pl.cellstart_f;
        S.sorted_f[i] = pl.sorted_f; S.out[i] = t.idx_out; S.kind[i] = pl.kind; S.dense_min = pl.dense_min;
        S.first[i] = blocks;
        const int tiles = pl.g.h8 * pl.g.w8;
        blocks += pl.kind == 0 ? cdiv(tiles, 4) : (pl.kind == 1 ? tiles : cdiv(t.h * t.w, 4));
        bytes += (double)B * ((double)t.h * t.w * K * 4.0 + (double)n_max * 16.0);
    }
    if (!merged) return DCF_OK;
    for (int i = nsites; i < DCF_MAX_KNN_SITES; ++i) {
        S.g[i] = S.g[0]; S.gf[i] = S.gf[0]; S.cellstart[i] = S.cellstart[0]; S.sorted[i] = S.sorted[0]; S.cellstart_f[i] = S.cellstart_f[0];
        S.sorted_f[i] = S.sorted_f[0]; S.out[i] = S.out[0]; S.kind[i] = S.kind[0];
    }
    for (int i = nsites; i <= DCF_MAX_KNN_SITES; ++i) S.first[i] = blocks;
#define KNN_MCASE(KK)                                                                                                                  \
    case KK:                                                                                                                           \
        DCF_LAUNCH_B("knn_search_sites", bytes, s, hipLaunchKernelGGL(k_knn_search_ms<KK>, dim3(blocks, B), dim3(256), 0, s, count_dev, n_max, S, rmax2)); \
        break;
    switch (K) {
        KNN_MCASE(1) KNN_MCASE(2) KNN_MCASE(3) KNN_MCASE(4) KNN_MCASE(5) KNN_MCASE(6) KNN_MCASE(7) KNN_MCASE(8)
    }
#undef KNN_MCASE
    return DCF_OK;
}

extern "C" size_t dcf_fusion_invert_workspace_bytes(int n_max, int nmaps)
{
    const size_t nscan = (size_t)nmaps * (n_max + 1);
    return sizeof(int) * (2 * nscan + (size_t)cdiv(nscan, CP_TILE) + 8);
}

extern "C" int dcf_fusion_invert(const dcf_knn_map *maps, int nmaps, int K, int n_max, int32_t *start, int32_t *ent_pix, int32_t *ent_pt,
                                 void *ws, dcf_stream_t stream)
{
    DCF_REQUIRE(maps && start && ent_pix && ent_pt && ws && K >= 1 && n_max >= 1, "dcf_fusion_invert: bad arguments");
    DCF_REQUIRE(nmaps >= 1 && nmaps <= DCF_MAX_KNN_MAPS, "dcf_fusion_invert: 1..%d maps per call", DCF_MAX_KNN_MAPS);
    InvMaps m;
    m.n = nmaps;
    int64_t tot = 0;
    for (int i = 0; i < DCF_MAX_KNN_MAPS; ++i) {
        m.idx[i] = nullptr; m.hw[i] = 1; m.w[i] = 1; m.first[i] = (int)tot;
        if (i < nmaps) {
            DCF_REQUIRE(maps[i].idx && maps[i].h >= 1 && maps[i].w >= 1 && maps[i].h < 65536 && maps[i].w < 65536,
                        "dcf_fusion_invert: map %d: null or pixel coordinates beyond 16 bits", i);
            m.idx[i] = maps[i].idx; m.hw[i] = maps[i].h * maps[i].w; m.w[i] = maps[i].w;
            tot += (int64_t)K * m.hw[i];
        }
    }
    m.first[DCF_MAX_KNN_MAPS] = (int)tot;
    for (int i = nmaps; i <= DCF_MAX_KNN_MAPS; ++i) m.first[i] = (int)tot;
    DCF_REQUIRE(tot < (1ll << 31), "dcf_fusion_invert: too many pairs");
    hipStream_t s = S(stream);
    const int total = (int)tot, nscan = nmaps * (n_max + 1), nsb = cdiv(nscan, CP_TILE);
    int *cnt = (int *)ws, *cursor = cnt + nscan, *blocksum = cursor + nscan, *totp = blocksum + nsb;
    DCF_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)nscan, s));
    DCF_LAUNCH_B("inv_hist", (double)total * 4.0, s, hipLaunchKernelGGL(k_inv_hist, dim3(cdiv(total, 256)), dim3(256), 0, s, m, n_max, cnt));
    DCF_LAUNCH("scan_blocksum", s, hipLaunchKernelGGL(k_scan_blocksum, dim3(nsb), dim3(CP_THREADS), 0, s, cnt, nscan, blocksum));
    DCF_LAUNCH("compact_scan", s, hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(CP_THREADS), 0, s, blocksum, nsb, totp));
    DCF_LAUNCH("scan_apply", s, hipLaunchKernelGGL(k_scan_apply, dim3(nsb), dim3(CP_THREADS), 0, s, cnt, nscan, blocksum, start, cursor));
    DCF_LAUNCH_B("inv_fill", (double)total * 12.0, s, hipLaunchKernelGGL(k_inv_fill, dim3(cdiv(total, 256)), dim3(256), 0, s, m, n_max, cursor, ent_pix, ent_pt));
    return DCF_OK;
}

#ifdef KNN_STAMP
extern "C" int dcf_knn_stamps_read(long long *dst, int *dims)
{
    dims[0] = KNN_STAMP_TILES; dims[1] = 16;
    DCF_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_knn_stamps), sizeof(long long) * KNN_STAMP_TILES * 16));
    return DCF_OK;
}
#endif
