"""GPU parity: geometry kernels (through the C ABI) against the oracle and the golden vectors."""
import numpy as np
import pytest
import torch

from _util import golden_cfg, load_golden, pkg
from oracle import geometry_ref

pytestmark = pytest.mark.gpu


def _spec(cfg):
    return pkg("ops").GridSpec(cfg)


@pytest.mark.parametrize("case", ["two", "five", "n1k", "n10k"])
def test_voxel_project_golden_bit_exact(case):
    ops, H = pkg("ops"), pkg("_hip")
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    g = _spec(cfg)
    pts = torch.from_numpy(z[case + "_pts"]).cuda()
    grid = ops.voxelize(pts, g.lim, g.aff, g.dims, H.VOXEL_COMPAT).cpu().numpy()
    want = np.zeros(grid.size, np.float32)
    want[z[case + "_vox_idx"]] = z[case + "_vox_val"]
    bad = np.flatnonzero(grid.reshape(-1).view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, "voxel grid differs at %d voxels, first %s" % (bad.size, bad[:5])
    uv, xyz, cnt, _ = ops.project_filter(pts, g.lim, z["crt"], cfg["image_height"], cfg["image_width"], n_out=cfg["max_num_pc"])
    n = int(cnt.item())
    assert n == int(z[case + "_n"])
    assert np.array_equal(uv.cpu().numpy()[:n].view(np.uint32), z[case + "_uv"].view(np.uint32))
    assert np.array_equal(xyz.cpu().numpy()[:n].view(np.uint32), z[case + "_xyz"].view(np.uint32))
    assert not uv.cpu().numpy()[n:].any() and not xyz.cpu().numpy()[n:].any()  # zero padding of :263-266
    # trunc'd voxel ids of the in-range points (debug BEV image path, :258, :269-272)
    pin, src, c2 = ops.range_filter(pts, g.lim)
    m = int(c2.item())
    assert m == z[case + "_ids"].shape[1]
    ref_in, ref_src = geometry_ref.range_filter(z[case + "_pts"], g.lim)
    assert np.array_equal(pin.cpu().numpy()[:m], ref_in) and np.array_equal(src.cpu().numpy()[:m], ref_src)


@pytest.mark.parametrize("case", ["two", "five", "n1k", "n10k"])
def test_occupancy_voxels_golden(case):
    """interpolate=False (data_import_carla.py:231-234) through the Dataset surface: exactly the reference's set voxels."""
    D, H = pkg("data_import_carla"), pkg("_hip")
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    geo = D.FrameGeometry(cfg, z["crt"])
    vox, _, _, cnt, _ = geo(torch.from_numpy(z[case + "_pts"]), voxel_mode=H.VOXEL_OCCUPANCY)
    got = vox.cpu().numpy().reshape(-1)
    assert set(np.unique(got)) <= {0.0, 1.0}
    assert np.array_equal(np.flatnonzero(got).astype(np.int32), z[case + "_occ_idx"])
    assert int(cnt.item()) == int(z[case + "_n"])


def test_voxel_workspace_returned_zero_and_reusable():
    ops, H = pkg("ops"), pkg("_hip")
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    g = _spec(cfg)
    Cz, L, W = g.dims
    ws = torch.zeros((2, Cz * L * W), dtype=torch.int32, device="cuda")
    pts = torch.from_numpy(z["n10k_pts"]).cuda()
    a = ops.voxelize(pts, g.lim, g.aff, g.dims, H.VOXEL_COMPAT, ws)
    assert int(ws.abs().sum().item()) == 0
    b = ops.voxelize(pts, g.lim, g.aff, g.dims, H.VOXEL_COMPAT, ws)
    assert torch.equal(a, b)


def test_voxel_accum_mode_and_empty_inputs():
    ops, H = pkg("ops"), pkg("_hip")
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    g = _spec(cfg)
    pts = z["n10k_pts"]
    pin, _ = geometry_ref.range_filter(pts, g.lim)
    ref = geometry_ref.voxelize(pin, g.aff, g.dims, "accum")
    got = ops.voxelize(torch.from_numpy(pts).cuda(), g.lim, g.aff, g.dims, H.VOXEL_ACCUM).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-5          # atomics: order differs, values agree
    assert abs(got.sum() - pin.shape[0]) < 1e-2    # trilinear weights sum to 1 per point
    empty = torch.zeros((0, 3), device="cuda")
    assert float(ops.voxelize(empty, g.lim, g.aff, g.dims, H.VOXEL_COMPAT).abs().sum()) == 0.0
    uv, xyz, cnt, _ = ops.project_filter(empty, g.lim, z["crt"], 480, 640, n_out=16)
    assert int(cnt.item()) == 0
    far = torch.full((100, 3), 500.0, device="cuda")   # everything out of range
    _, _, cnt, _ = ops.project_filter(far, g.lim, z["crt"], 480, 640)
    assert int(cnt.item()) == 0


def test_full_size_100k_points_vs_oracle():
    """BASELINE full size: 100k-point cloud on the KITTI-scale grid, bit-exact against the C oracle."""
    ops, H, det = pkg("ops"), pkg("_hip"), pkg("detfill")
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    cfg.update(dict(voxel_length=704, voxel_width=800, lidar_x_max=70.4, lidar_y_min=-40.0, lidar_y_max=40.0,
                    image_height=375, image_width=1242, max_num_pc=100000))
    g = _spec(cfg)
    crt = pkg("calib").kitti_like_crt()
    pts = det.synthetic_points(100000, (0.0, 70.4, -40.0, 40.0, -2.4, 0.8), seed=5)
    ref_grid, ref_pc, ref_uv, ref_n, _ = geometry_ref.voxelization_projection(pts, cfg, crt, proj_mode="correct")
    d = torch.from_numpy(pts).cuda()
    grid = ops.voxelize(d, g.lim, g.aff, g.dims, H.VOXEL_COMPAT)
    assert np.array_equal(grid.cpu().numpy().view(np.uint32), ref_grid.view(np.uint32))
    uv, xyz, cnt, _ = ops.project_filter(d, g.lim, crt, 1242, 375, H.PROJ_CORRECT)
    n = int(cnt.item())
    assert n == ref_n and n > 1000
    assert np.array_equal(uv.cpu().numpy()[:n].view(np.uint32), ref_uv[:n].view(np.uint32))
    assert np.array_equal(xyz.cpu().numpy()[:n].view(np.uint32), ref_pc[:n].view(np.uint32))


@pytest.mark.parametrize("K", [1, 3, 5])
@pytest.mark.parametrize("stride", [2, 4, 16])
def test_knn_bit_exact_vs_bruteforce(K, stride):
    ops = pkg("ops")
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    g = _spec(cfg)
    xyz = z["n10k_xyz"]
    h, w = cfg["voxel_length"] // stride, cfg["voxel_width"] // stride
    if stride == 2:   # keep the brute-force oracle in seconds: crop the BEV to a window near the sensor
        h, w = 96, 128
    ref = geometry_ref.knn_bev(xyz, K, h, w, stride, g.aff)
    d = torch.from_numpy(xyz).cuda()
    cnt = torch.tensor([xyz.shape[0]], dtype=torch.int32, device="cuda")
    got = ops.knn_bev(d, cnt, K, h, w, stride, g.aff).cpu().numpy()
    bad = np.argwhere(got != ref)
    assert bad.shape[0] == 0, "KNN differs at %d entries, first %s got %s ref %s" % (
        bad.shape[0], bad[:3].tolist(), got[tuple(bad[0])] if bad.size else None, ref[tuple(bad[0])] if bad.size else None)


def test_knn_edge_cases():
    ops = pkg("ops")
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    g = _spec(cfg)
    h, w, s, K = 48, 32, 8, 3
    xyz = z["n1k_xyz"]
    d = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
    # count < K  -> -1 padding ; count == 0 -> all -1 ; rows past count are ignored
    for n in (0, 1, 2):
        cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
        got = ops.knn_bev(d, cnt, K, h, w, s, g.aff).cpu().numpy()
        assert np.array_equal(got, geometry_ref.knn_bev(xyz[:n], K, h, w, s, g.aff))
    # radius cut
    cnt = torch.tensor([xyz.shape[0]], dtype=torch.int32, device="cuda")
    for rmax in (0.5, 3.0, 40.0):
        got = ops.knn_bev(d, cnt, K, h, w, s, g.aff, rmax=rmax).cpu().numpy()
        assert np.array_equal(got, geometry_ref.knn_bev(xyz, K, h, w, s, g.aff, rmax=rmax)), rmax
    # duplicates: identical points must come out in index order (tie-break on index)
    dup = np.repeat(xyz[:5], 4, axis=0)
    cnt = torch.tensor([dup.shape[0]], dtype=torch.int32, device="cuda")
    got = ops.knn_bev(torch.from_numpy(dup).cuda(), cnt, 5, h, w, s, g.aff).cpu().numpy()
    assert np.array_equal(got, geometry_ref.knn_bev(dup, 5, h, w, s, g.aff))
    # a single far cluster: every pixel must still find it (coarse-ring phase)
    far = (xyz[:8] * 0 + np.array([68.0, 28.0, -1.0], np.float32) + np.arange(8, dtype=np.float32)[:, None] * 0.01).astype(np.float32)
    cnt = torch.tensor([8], dtype=torch.int32, device="cuda")
    got = ops.knn_bev(torch.from_numpy(far).cuda(), cnt, 3, h, w, s, g.aff).cpu().numpy()
    assert np.array_equal(got, geometry_ref.knn_bev(far, 3, h, w, s, g.aff))


@pytest.mark.parametrize("npts,hw,K", [(120000, (375, 1242), 5), (300000, (1080, 1920), 3)])
def test_cfg4_cfg5_geometry_scale(npts, hw, K):
    """BASELINE configs[3]/[4] input sizes (120k pts + 1242x375, K=5; 300k pts + 1920x1080): voxel grid, projection
    and compaction bit-exact against the C oracle; KNN bit-exact on a BEV window (brute force kept to seconds)."""
    ops, H, det, calib = pkg("ops"), pkg("_hip"), pkg("detfill"), pkg("calib")
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    cfg.update(dict(voxel_length=704, voxel_width=800, lidar_x_max=70.4, lidar_y_min=-40.0, lidar_y_max=40.0,
                    image_height=hw[0], image_width=hw[1], max_num_pc=npts))
    g = _spec(cfg)
    crt = calib.kitti_like_crt() if hw[0] == 375 else calib.hd_crt()
    pts = det.synthetic_points(npts, (0.0, 70.4, -40.0, 40.0, -2.4, 0.8), seed=11)
    ref_grid, ref_pc, ref_uv, ref_n, _ = geometry_ref.voxelization_projection(pts, cfg, crt, proj_mode="correct")
    d = torch.from_numpy(pts).cuda()
    grid = ops.voxelize(d, g.lim, g.aff, g.dims, H.VOXEL_COMPAT)
    assert np.array_equal(grid.cpu().numpy().view(np.uint32), ref_grid.view(np.uint32))
    uv, xyz, cnt, _ = ops.project_filter(d, g.lim, crt, hw[1], hw[0], H.PROJ_CORRECT)
    n = int(cnt.item())
    assert n == ref_n and n > 10000
    assert np.array_equal(uv.cpu().numpy()[:n].view(np.uint32), ref_uv[:n].view(np.uint32))
    assert np.array_equal(xyz.cpu().numpy()[:n].view(np.uint32), ref_pc[:n].view(np.uint32))
    # KNN: full site on the GPU, a 24x200-pixel strip checked against the brute-force oracle
    s = 4
    h, w = 704 // s, 800 // s
    got = ops.knn_bev(xyz, cnt, K, h, w, s, g.aff).cpu().numpy()
    ref = geometry_ref.knn_bev(ref_pc[:n], K, 24, w, s, g.aff)       # first 24 BEV rows
    assert np.array_equal(got[:, :24, :], ref)
    # size-independent property on the whole site: every index valid, neighbours sorted by (d2, index)
    assert got.min() >= 0 and got.max() < n
    p = ref_pc[:n]
    X = ((np.arange(h, dtype=np.float32) + np.float32(0.5)) * np.float32(s) - g.aff[1]) / g.aff[0]
    Y = ((np.arange(w, dtype=np.float32) + np.float32(0.5)) * np.float32(s) - g.aff[3]) / g.aff[2]
    d2 = []
    for k in range(K):
        dx = p[got[k], 0] - X[:, None]
        dy = p[got[k], 1] - Y[None, :]
        d2.append((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32))
    for k in range(K - 1):
        assert ((d2[k] < d2[k + 1]) | ((d2[k] == d2[k + 1]) & (got[k] < got[k + 1]))).all()


def test_voxelize_batch_equals_per_frame():
    """dcf_voxelize_batch (frames side by side in one launch per owner round) == one dcf_voxelize per frame, bit for bit,
    for frames of different sizes (including an empty one); the owner workspace comes back zero."""
    ops, H, det = pkg("ops"), pkg("_hip"), pkg("detfill")
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    g = _spec(cfg)
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    frames = [torch.from_numpy(det.synthetic_points(n, lim6, 70 + i)).cuda() if n else torch.zeros(0, 3, device="cuda")
              for i, n in enumerate((5000, 0, 12345))]
    Cz, L, W = g.dims
    out = torch.full((3, Cz, L, W), float("nan"), device="cuda")
    owner = torch.zeros((3, 2, Cz * L * W), dtype=torch.int32, device="cuda")
    ops.voxelize_batch(frames, g.lim, g.aff, g.dims, owner, out)
    assert int(owner.abs().max()) == 0
    for b, p in enumerate(frames):
        ref = ops.voxelize(p, g.lim, g.aff, g.dims, H.VOXEL_COMPAT)
        assert torch.equal(out[b].view(torch.int32), ref.view(torch.int32))
        lit = ops.voxelize(p, g.lim, g.aff, g.dims, H.VOXEL_COMPAT_ROUNDS)          # the literal nine-round formulation
        assert torch.equal(out[b].view(torch.int32), lit.view(torch.int32))


@pytest.mark.parametrize("case", ["dense", "clustered", "one_cell", "edges", "tight_limits"])
def test_voxel_cell_formulation_equals_nine_rounds(case):
    """The three-launch compat voxeliser (last point of every CELL decides all eight passes; each cell winner evaluates the
    ordered sums of its corners) against the literal nine rounds (claim pass c / resolve pass c-1), bit for bit, on clouds
    with many points per cell, shared voxels between neighbouring cells and points on the range limits."""
    ops, H, det = pkg("ops"), pkg("_hip"), pkg("detfill")
    # tight_limits: the tiny model grid leaves less than one cell of margin in y, so a corner index runs one past the end of
    # a row -- on the flattened grid that is the first voxel of the next row, in both formulations and in the CPU restatement
    cfg = golden_cfg(load_golden("model_tiny.npz" if case == "tight_limits" else "geometry_carla.npz"))
    g = _spec(cfg)
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    if case in ("dense", "tight_limits"):
        p = det.synthetic_points(60000, lim6, 5)
    elif case == "clustered":                       # 40 k points inside a 2 m cube: dozens of points per cell
        p = det.synthetic_points(40000, lim6, 6)
        p = (p - p.mean(0)) * 0.02 + np.array([20.0, 3.0, -1.0], dtype=np.float32)
    elif case == "one_cell":
        p = np.tile(np.array([[10.03, 1.01, -0.52]], dtype=np.float32), (500, 1)) + det.uniform((500, 3), 9, 0.0, 0.02)
    else:                                           # points at / next to the limits (some filtered, some in the last cells)
        u = det.uniform((4000, 3), 12, 0.0, 1.0)
        lo = np.array([lim6[0], lim6[2], lim6[4]], dtype=np.float32)
        hi = np.array([lim6[1], lim6[3], lim6[5]], dtype=np.float32)
        p = np.where(u < 0.5, lo + u * 0.6, hi - (1.0 - u) * 0.6).astype(np.float32)
    pts = torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)).cuda()
    a = ops.voxelize(pts, g.lim, g.aff, g.dims, H.VOXEL_COMPAT)
    b = ops.voxelize(pts, g.lim, g.aff, g.dims, H.VOXEL_COMPAT_ROUNDS)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and float(a.sum()) > 0


@pytest.mark.parametrize("dtype", [1, 2, 0])
def test_voxelize_batch_nhwc_equals_transposed_grid(dtype):
    """dcf_voxelize_batch_nhwc writes the engine's input image directly; it must equal dcf_nchw_to_nhwc of the fp32 grids."""
    ops, det = pkg("ops"), pkg("detfill")
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    g = _spec(cfg)
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    frames = [torch.from_numpy(det.synthetic_points(n, lim6, 80 + i)).cuda() for i, n in enumerate((9000, 30000))]
    Cz, L, W = g.dims
    owner = torch.zeros((2, 2, Cz * L * W), dtype=torch.int32, device="cuda")
    grids = torch.empty((2, Cz, L, W), device="cuda")
    ops.voxelize_batch(frames, g.lim, g.aff, g.dims, owner, grids)
    want = ops.nchw_to_nhwc(grids, dtype)
    tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dtype]
    got = torch.full((2, L, W, Cz), 7.0, dtype=tdt, device="cuda")
    ops.voxelize_batch_nhwc(dtype, frames, g.lim, g.aff, g.dims, owner, got)
    assert int(owner.abs().max()) == 0
    assert torch.equal(got.view(torch.int32 if dtype == 0 else torch.int16), want.view(torch.int32 if dtype == 0 else torch.int16))


def _cfg2_cloud(seed=5, npts=100000):
    """cfg2 geometry (BASELINE.json configs[1]): 100 k points, KITTI-like calibration -> the ~40 k in-frustum points the
    KNN really sees, as the C oracle's projection produces them."""
    det, calib = pkg("detfill"), pkg("calib")
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    cfg.update(dict(voxel_length=704, voxel_width=800, lidar_x_max=70.4, lidar_y_min=-40.0, lidar_y_max=40.0,
                    image_height=375, image_width=1242, max_num_pc=npts))
    g = _spec(cfg)
    pts = det.synthetic_points(npts, (0.0, 70.4, -40.0, 40.0, -2.4, 0.8), seed=seed)
    _, pc, _, n, _ = geometry_ref.voxelization_projection(pts, cfg, calib.kitti_like_crt(), proj_mode="correct")
    return g, np.ascontiguousarray(pc[:n])


def _knn_check_pixels(got, xyz, K, pi, pj, stride, aff, rmax=None):
    ref = geometry_ref.knn_pixels(xyz, K, pi, pj, stride, aff, rmax)            # [npix, K]
    mine = got[:, pi, pj].T
    bad = np.flatnonzero((mine != ref).any(1))
    assert bad.size == 0, "KNN differs at %d of %d pixels, first (%d,%d): got %s ref %s" % (
        bad.size, pi.size, pi[bad[0]], pj[bad[0]], mine[bad[0]].tolist(), ref[bad[0]].tolist())


@pytest.mark.parametrize("K", [3, 5])
@pytest.mark.parametrize("stride", [2, 4])
def test_knn_cfg2_whole_site_sampled_bruteforce(K, stride):
    """The tile kernel (k_knn_search: sites > 20 000 px, the one that produces cfg2's stride-2 / stride-4 maps) against
    the brute-force oracle over the WHOLE site: 4096 uniformly random pixels, every pixel of the last 24 rows (the far
    field, beyond the point cloud's dense region), the first rows, and the left / right border columns (outside the
    camera frustum: the coarse 8x8-block ring phase with bounding-box pruning)."""
    ops = pkg("ops")
    g, xyz = _cfg2_cloud()
    n = xyz.shape[0]
    assert 20000 < n < 80000
    h, w = 704 // stride, 800 // stride
    assert h * w > 20000
    d = torch.from_numpy(xyz).cuda()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    got = ops.knn_bev(d, cnt, K, h, w, stride, g.aff).cpu().numpy()
    assert got.min() >= 0 and got.max() < n
    rng = np.random.default_rng(1234 + K * 10 + stride)
    pi = rng.integers(0, h, 4096).astype(np.int32)
    pj = rng.integers(0, w, 4096).astype(np.int32)
    _knn_check_pixels(got, xyz, K, pi, pj, stride, g.aff)
    ii, jj = np.meshgrid(np.arange(h - 24, h), np.arange(w), indexing="ij")     # far field: last 24 rows, all columns
    _knn_check_pixels(got, xyz, K, ii.ravel().astype(np.int32), jj.ravel().astype(np.int32), stride, g.aff)
    ii, jj = np.meshgrid(np.arange(0, h, 3), np.concatenate([np.arange(0, 8), np.arange(w - 8, w)]), indexing="ij")
    _knn_check_pixels(got, xyz, K, ii.ravel().astype(np.int32), jj.ravel().astype(np.int32), stride, g.aff)


@pytest.mark.parametrize("stride,K", [(2, 3), (4, 5), (8, 3)])
def test_knn_tile_kernel_equals_wave_kernel_full_site(stride, K):
    """k_knn_search (one wave per 8x8 tile, window + coarse rings) and k_knn_search_wave (one wave per pixel, lanes split
    the candidates) are two implementations of the same exact (d2, index) order: whole-site equality at cfg2 size, in both
    directions of the dispatch threshold (a fine site forced onto the wave kernel, a coarse one onto the tile kernel)."""
    ops = pkg("ops")
    g, xyz = _cfg2_cloud(seed=9)
    n = xyz.shape[0]
    h, w = 704 // stride, 800 // stride
    d = torch.from_numpy(xyz).cuda()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    H = pkg("_hip")
    try:
        H.set_option("KNN_KERNEL", "tile")
        a = ops.knn_bev(d, cnt, K, h, w, stride, g.aff)
        H.set_option("KNN_KERNEL", "wave")
        b = ops.knn_bev(d, cnt, K, h, w, stride, g.aff)
        H.set_option("KNN_KERNEL", None)
        c = ops.knn_bev(d, cnt, K, h, w, stride, g.aff)
        assert torch.equal(a, b) and torch.equal(a, c)
        # ... and with a radius cut that leaves pixels with fewer than K neighbours
        H.set_option("KNN_KERNEL", "tile")
        a = ops.knn_bev(d, cnt, K, h, w, stride, g.aff, rmax=1.5)
        H.set_option("KNN_KERNEL", "wave")
        b = ops.knn_bev(d, cnt, K, h, w, stride, g.aff, rmax=1.5)
    finally:
        H.set_option("KNN_KERNEL", None)
    assert torch.equal(a, b) and int((a < 0).sum()) > 0


@pytest.mark.parametrize("where", ["far_corner", "near_corner", "two_clusters"])
def test_knn_tile_kernel_far_cluster_full_site(where):
    """A fine site (> 20 000 px, tile kernel) whose only points are one or two tiny clusters: almost every tile finds its
    12x12-cell window empty and reaches the cluster through many coarse 8x8-block rings.  Whole site against brute force
    (few points: the full-site brute force is cheap)."""
    ops = pkg("ops")
    g, _ = _cfg2_cloud()
    stride, K = 4, 3
    h, w = 704 // stride, 800 // stride
    base = {"far_corner": [(69.7, 39.2)], "near_corner": [(0.3, -39.6)], "two_clusters": [(69.9, -39.9), (35.0, 3.0)]}[where]
    rows = []
    for cx, cy in base:
        for t in range(7):
            rows.append([cx + 0.013 * t, cy - 0.007 * t, -1.0 + 0.1 * t])
    xyz = np.asarray(rows, dtype=np.float32)
    d = torch.from_numpy(xyz).cuda()
    cnt = torch.tensor([xyz.shape[0]], dtype=torch.int32, device="cuda")
    got = ops.knn_bev(d, cnt, K, h, w, stride, g.aff).cpu().numpy()
    assert np.array_equal(got, geometry_ref.knn_bev(xyz, K, h, w, stride, g.aff))
    got = ops.knn_bev(d, cnt, K, h, w, stride, g.aff, rmax=30.0).cpu().numpy()
    assert np.array_equal(got, geometry_ref.knn_bev(xyz, K, h, w, stride, g.aff, rmax=30.0))


@pytest.mark.parametrize("cloud", ["cfg2", "far_corner", "two_clusters", "one_point"])
def test_knn_tile_kernel_waves_per_tile_change_nothing(cloud):
    """Round 4: k_knn_search runs a tile with one, two or four waves that share the candidates (every wave takes every NW-th
    point of the same cells and rings) and merge their K-best lists through LDS where the pixel's true K-th decides something;
    the lists hold 64-bit (d2 bits, index) keys and insertion is a branch-free compare-exchange chain.  Whole-site, bit-for-bit
    equality of the three with each other and with the per-pixel wave kernel, K = 1 .. 8, with and without radius cuts."""
    ops, H = pkg("ops"), pkg("_hip")
    g, xyz = _cfg2_cloud(seed=21)
    if cloud != "cfg2":
        base = {"far_corner": [(69.7, 39.2)], "two_clusters": [(69.9, -39.9), (35.0, 3.0)], "one_point": [(12.3, -20.1)]}[cloud]
        rows = [[cx + 0.013 * t, cy - 0.007 * t, -1.0 + 0.1 * t] for cx, cy in base for t in range(1 if cloud == "one_point" else 7)]
        xyz = np.asarray(rows, dtype=np.float32)
    n = xyz.shape[0]
    d = torch.from_numpy(xyz).cuda()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    try:
        for rmax in (None, 30.0, 1.5):
            for stride, K in ((2, 3), (4, 5), (4, 1), (4, 8), (8, 2)):
                h, w = 704 // stride, 800 // stride
                H.set_option("KNN_KERNEL", "wave")
                ref = ops.knn_bev(d, cnt, K, h, w, stride, g.aff, rmax)
                H.set_option("KNN_KERNEL", "tile")
                for tw in ("1", "2", "4", None):
                    H.set_option("KNN_TILE_WAVES", tw)
                    got = ops.knn_bev(d, cnt, K, h, w, stride, g.aff, rmax)
                    assert torch.equal(got, ref), (rmax, stride, K, tw)
            if rmax == 1.5:
                assert int((ref < 0).sum()) > 0
    finally:
        H.set_option("KNN_TILE_WAVES", None)
        H.set_option("KNN_KERNEL", None)


@pytest.mark.parametrize("stride,K", [(2, 3), (8, 5), (16, 3)])
def test_knn_batch_equals_per_frame(stride, K):
    """dcf_knn_bev_batch (every phase once for the batch, grid.y = frame) == dcf_knn_bev frame by frame, bit for bit: frames with
    different valid counts (one of them empty), tile kernel and wave kernel sites."""
    ops, det = pkg("ops"), pkg("detfill")
    g, xyz = _cfg2_cloud(seed=13)
    n_max = 40960
    h, w = 704 // stride, 800 // stride
    B = 3
    pts = torch.zeros((B, n_max, 3), device="cuda")
    counts = [xyz.shape[0], 0, 1234]
    _, xyz2 = _cfg2_cloud(seed=14)
    pts[0, :counts[0]] = torch.from_numpy(xyz).cuda()
    pts[2, :counts[2]] = torch.from_numpy(xyz2[:counts[2]]).cuda()
    cnt = torch.tensor(counts, dtype=torch.int32, device="cuda")
    got = ops.knn_bev_batch(pts, cnt, K, h, w, stride, g.aff)
    for b in range(B):
        ref = ops.knn_bev(pts[b], cnt[b:b + 1], K, h, w, stride, g.aff)
        assert torch.equal(got[b], ref), "frame %d" % b
    assert int((got[1] != -1).sum()) == 0
    got2 = ops.knn_bev_batch(pts, cnt, K, h, w, stride, g.aff, rmax=2.0)
    for b in range(B):
        assert torch.equal(got2[b], ops.knn_bev(pts[b], cnt[b:b + 1], K, h, w, stride, g.aff, rmax=2.0))


def _coarse_on_fine(ops, d, cnt, K, stride, aff, rmax=None, fine_stride=2):
    """Maps of one coarse site two ways: its own cell sort alone (dcf_knn_bev_batch: wave / tile kernel) and with the dense
    regions served from the finer site's cells (dcf_knn_bev_batch_shared: k_knn_search_fine)."""
    B, n_max = d.shape[0], d.shape[1]
    fh, fw = 704 // fine_stride, 800 // fine_stride
    h, w = 704 // stride, 800 // stride
    ws = torch.empty((B, ops.knn_ws_stride(n_max, fh, fw)), dtype=torch.uint8, device="cuda")
    ops.knn_bev_batch(d, cnt, K, fh, fw, fine_stride, aff, rmax, ws=ws)
    own = ops.knn_bev_batch(d, cnt, K, h, w, stride, aff, rmax)
    shared = ops.knn_bev_batch_shared(d, cnt, K, h, w, stride, (fh, fw, fine_stride), ws, aff, rmax)
    return own, shared


@pytest.mark.parametrize("stride,K", [(8, 3), (16, 3), (16, 5), (4, 1), (8, 8)])
def test_knn_coarse_site_on_fine_cells_equals_own_sort(stride, K):
    """dcf_knn_bev_batch_shared (round 3: coarse sites searched on the stride-2 site's cells, k_knn_search_fine: windows of fine
    cells, then block rings, exact open-edge termination) against the site's own sort + wave / tile kernel: whole-site equality
    on two cfg2 frames of different density, with and without a radius cut."""
    ops = pkg("ops")
    g, a = _cfg2_cloud(seed=9)
    _, b = _cfg2_cloud(seed=10, npts=30000)
    n_max = max(a.shape[0], b.shape[0])
    d = torch.zeros(2, n_max, 3)
    d[0, :a.shape[0]] = torch.from_numpy(a); d[1, :b.shape[0]] = torch.from_numpy(b)
    d = d.cuda()
    cnt = torch.tensor([a.shape[0], b.shape[0]], dtype=torch.int32, device="cuda")
    own, shared = _coarse_on_fine(ops, d, cnt, K, stride, g.aff)
    assert torch.equal(own, shared) and int(own.min()) >= 0
    own, shared = _coarse_on_fine(ops, d, cnt, K, stride, g.aff, rmax=1.5)
    assert torch.equal(own, shared) and int((own < 0).sum()) > 0
    # brute-force spot check of the shared result itself (sampled pixels of frame 0)
    h, w = 704 // stride, 800 // stride
    rng = np.random.default_rng(77)
    pi = rng.integers(0, h, 1500).astype(np.int32); pj = rng.integers(0, w, 1500).astype(np.int32)
    _knn_check_pixels(_coarse_on_fine(ops, d, cnt, K, stride, g.aff)[1][0].cpu().numpy(), a, K, pi, pj, stride, g.aff)


@pytest.mark.parametrize("where", ["far_corner", "near_corner", "two_clusters", "few", "none", "on_borders"])
def test_knn_coarse_site_on_fine_cells_sparse_clouds(where):
    """The block-ring phase and the closed-edge logic of k_knn_search_fine: clusters far from most pixels, fewer points than K,
    no points at all, points sitting on the grid's border cells."""
    ops = pkg("ops")
    g, _ = _cfg2_cloud()
    rng = np.random.default_rng(5)
    if where == "far_corner":
        xyz = np.stack([rng.uniform(66.0, 70.3, 400), rng.uniform(36.0, 39.9, 400), rng.uniform(-1, 1, 400)], 1)
    elif where == "near_corner":
        xyz = np.stack([rng.uniform(0.0, 1.5, 300), rng.uniform(-39.9, -38.0, 300), rng.uniform(-1, 1, 300)], 1)
    elif where == "two_clusters":
        a = np.stack([rng.uniform(10.0, 11.0, 200), rng.uniform(-30.0, -29.0, 200), rng.uniform(-1, 1, 200)], 1)
        b = np.stack([rng.uniform(60.0, 61.0, 5), rng.uniform(30.0, 31.0, 5), rng.uniform(-1, 1, 5)], 1)
        xyz = np.concatenate([a, b])
    elif where == "few":
        xyz = np.array([[35.0, 0.0, 0.0], [35.05, 0.01, 0.0]])
    elif where == "none":
        xyz = np.zeros((0, 3))
    else:
        t = np.linspace(0.0, 1.0, 150)
        xyz = np.concatenate([np.stack([np.full(150, 0.001), -39.99 + 79.98 * t, t], 1), np.stack([70.39 * t, np.full(150, 39.99), t], 1),
                              np.stack([np.full(150, 70.39), -39.99 + 79.98 * t, t], 1)])
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    d = torch.zeros(1, max(n, 4), 3)
    d[0, :n] = torch.from_numpy(xyz)
    d = d.cuda()
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    for stride in (8, 16):
        for K in (3, 5):
            own, shared = _coarse_on_fine(ops, d, cnt, K, stride, g.aff)
            assert torch.equal(own, shared), (where, stride, K)
            own, shared = _coarse_on_fine(ops, d, cnt, K, stride, g.aff, rmax=3.0)
            assert torch.equal(own, shared), (where, stride, K, "rmax")
    if n >= 1:
        h, w = 704 // 16, 800 // 16
        ii, jj = np.meshgrid(np.arange(0, h, 3), np.arange(0, w, 3), indexing="ij")
        _knn_check_pixels(_coarse_on_fine(ops, d, cnt, 3, 16, g.aff)[1][0].cpu().numpy(), xyz, 3, ii.ravel().astype(np.int32), jj.ravel().astype(np.int32), 16, g.aff)


@pytest.mark.parametrize("K,B,rmax", [(3, 2, None), (5, 3, 2.0), (1, 1, None)])
def test_knn_all_sites_in_one_call_equal_the_per_site_calls(K, B, rmax):
    """dcf_knn_bev_sites (the four sites' cell sorts in one launch per phase, then the searches) against dcf_knn_bev_batch for the
    two fine sites and dcf_knn_bev_batch_shared for the two coarse ones: every map bit for bit, on frames of different density
    (one of them empty when B = 3), with and without a radius cut; a second call on the same workspaces gives the same maps."""
    ops, H = pkg("ops"), pkg("_hip")
    g, a = _cfg2_cloud(seed=21)
    clouds = [a, _cfg2_cloud(seed=22, npts=25000)[1], np.zeros((0, 3), np.float32)][:B]
    n_max = max(c.shape[0] for c in clouds)
    d = torch.zeros(B, n_max, 3)
    for b, c in enumerate(clouds):
        d[b, :c.shape[0]] = torch.from_numpy(c)
    d = d.cuda()
    cnt = torch.tensor([c.shape[0] for c in clouds], dtype=torch.int32, device="cuda")
    dims = [(704 // s, 800 // s, s) for s in (2, 4, 8, 16)]
    want = []
    ws0 = torch.empty((B, ops.knn_ws_stride(n_max, *dims[0][:2])), dtype=torch.uint8, device="cuda")
    for i, (h, w, s) in enumerate(dims):
        if i == 0:
            want.append(ops.knn_bev_batch(d, cnt, K, h, w, s, g.aff, rmax, ws=ws0))
        elif h * w > 20000:
            want.append(ops.knn_bev_batch(d, cnt, K, h, w, s, g.aff, rmax))
        else:
            want.append(ops.knn_bev_batch_shared(d, cnt, K, h, w, s, dims[0], ws0, g.aff, rmax))
    sites = []
    for i, (h, w, s) in enumerate(dims):
        sites.append((h, w, s, 0 if (i > 0 and h * w <= 20000) else -1,
                      torch.empty((B, ops.knn_ws_stride(n_max, h, w)), dtype=torch.uint8, device="cuda"),
                      torch.full((B, K, h, w), -7, dtype=torch.int32, device="cuda")))
    got = ops.knn_bev_sites(d, cnt, K, sites, g.aff, rmax)
    for i in range(4):
        assert torch.equal(got[i], want[i]), "site %d" % i
    again = [t.clone() for t in got]
    ops.knn_bev_sites(d, cnt, K, sites, g.aff, rmax)
    for i in range(4):
        assert torch.equal(sites[i][5], again[i])
    # round 4: the four searches are ONE launch (k_knn_search_ms: every site runs the kernel body it would run alone).  Every
    # body of that launch -- tile kernel with one / four waves per tile, wave kernel, fine-cell kernel -- and the per-site
    # launches of round 3 (KNN_MERGED_SEARCH=0) give the same bits.
    try:
        for opt, val in (("KNN_TILE_WAVES", "1"), ("KNN_TILE_WAVES", "4"), ("KNN_KERNEL", "wave"), ("KNN_KERNEL", "tile"), ("KNN_MERGED_SEARCH", "0")):
            H.set_option(opt, val)
            for t in sites:
                t[5].fill_(-7)
            ops.knn_bev_sites(d, cnt, K, sites, g.aff, rmax)
            H.set_option(opt, None)
            for i in range(4):
                assert torch.equal(sites[i][5], again[i]), (opt, val, i)
    finally:
        for opt in ("KNN_TILE_WAVES", "KNN_KERNEL", "KNN_MERGED_SEARCH"):
            H.set_option(opt, None)
    with pytest.raises(H.DcfError):             # `fine` must name an earlier site
        bad = list(sites)
        bad[1] = bad[1][:3] + (2,) + bad[1][4:]
        ops.knn_bev_sites(d, cnt, K, bad, g.aff, rmax)


@pytest.mark.parametrize("mode", [0, 1])
def test_project_filter_batch_equals_per_frame(mode):
    """dcf_project_filter_batch (round 5: count / scan / scatter once for all frames of a batch, blockIdx.y = frame) against one
    dcf_project_filter per frame, bit for bit: frames of different sizes (one empty, one smaller than a compaction tile), every
    frame with its own projection matrix, compat and correct bounds; rows past a frame's count stay untouched (zero)."""
    ops, H, det, calib = pkg("ops"), pkg("_hip"), pkg("detfill"), pkg("calib")
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    g = _spec(cfg)
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    sizes = (30000, 0, 700, 12345, 100000)
    frames = [torch.from_numpy(det.synthetic_points(n, lim6, 90 + i)).cuda() if n else torch.zeros(0, 3, device="cuda") for i, n in enumerate(sizes)]
    base = np.asarray(load_golden("geometry_carla.npz")["crt"], dtype=np.float32)
    crts = [base * np.float32(1.0 + 0.01 * b) for b in range(len(sizes))]
    rows = 100000
    B = len(sizes)
    uv = torch.zeros((B, rows, 2), device="cuda"); xyz = torch.zeros((B, rows, 3), device="cuda")
    cnt = torch.full((B,), -7, dtype=torch.int32, device="cuda")
    ops.project_filter_batch(frames, g.lim, np.stack(crts, 0), float(cfg["image_height"]), float(cfg["image_width"]), mode, uv, xyz, cnt)
    for b, p in enumerate(frames):
        if p.shape[0] == 0:
            assert int(cnt[b]) == 0 and not uv[b].any() and not xyz[b].any()
            continue
        uv1, xyz1, c1, _ = ops.project_filter(p, g.lim, crts[b], cfg["image_height"], cfg["image_width"], mode=mode, n_out=rows)
        assert int(cnt[b]) == int(c1.item()) and (int(c1.item()) > 0 or mode == 1)       # (the CARLA matrix keeps nothing under the corrected bounds)
        assert torch.equal(uv[b].view(torch.int32), uv1.view(torch.int32)) and torch.equal(xyz[b].view(torch.int32), xyz1.view(torch.int32))
