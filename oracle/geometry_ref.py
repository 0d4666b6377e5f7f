"""TEST INFRASTRUCTURE -- ctypes front-end of oracle/dcf_oracle.c (the CPU checker).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Parity status is stated in dcf_oracle.c's header.

Every function cites the reference lines it follows; grid constants restate
data_import_carla.py:35-43 (the int() truncations are part of the contract).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C checker (gcc).  Called by __graft_entry__.build()."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libdcf_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdcf_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        lp = ctypes.POINTER(ctypes.c_int64)
        L.dcf_oracle_range_filter.restype = ctypes.c_int
        L.dcf_oracle_range_filter.argtypes = [fp, ctypes.c_int, fp, fp, ip]
        L.dcf_oracle_voxelize.restype = None
        L.dcf_oracle_voxelize.argtypes = [fp, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, fp, lp]
        L.dcf_oracle_project.restype = ctypes.c_int
        L.dcf_oracle_project.argtypes = [fp, ctypes.c_int, fp, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_int, fp, fp, ip]
        L.dcf_oracle_knn_bev.restype = None
        L.dcf_oracle_knn_bev.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, ctypes.c_float, ip]
        L.dcf_oracle_knn_pixels.restype = None
        L.dcf_oracle_knn_pixels.argtypes = [fp, ctypes.c_int, ctypes.c_int, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ip]
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def grid_constants(cfg):
    """Restates data_import_carla.py:35-40 and the filter thresholds of :215-226.

    Returns dict(lim=float32[6], aff=float32[6] (sx,ox,sy,oy,sz,oz), dims=(Cz,L,W)).
    Thresholds are Python doubles rounded once to fp32 (torch compares the fp32
    tensor against the scalar in fp32).
    """
    L, W, Cz = cfg["voxel_length"], cfg["voxel_width"], cfg["voxel_channel"]
    xs = int(L / (cfg["lidar_x_max"] - cfg["lidar_x_min"]))
    ys = int(W / (cfg["lidar_y_max"] - cfg["lidar_y_min"]))
    zs = int(Cz / (cfg["lidar_z_max"] - cfg["lidar_z_min"]))
    xo = int(-cfg["lidar_x_min"] * xs)
    yo = int(-cfg["lidar_y_min"] * ys)
    zo = int(-cfg["lidar_z_min"] * zs)
    d = cfg["delta"]
    lim = np.array([cfg["lidar_x_min"], cfg["lidar_x_max"] - d, cfg["lidar_y_min"], cfg["lidar_y_max"] - d,
                    cfg["lidar_z_min"], cfg["lidar_z_max"] - d], dtype=np.float64).astype(np.float32)
    aff = np.array([xs, xo, ys, yo, zs, zo], dtype=np.float32)
    return {"lim": lim, "aff": aff, "dims": (Cz, L, W)}


def range_filter(pts, lim):
    """data_import_carla.py:215-229.  pts [N,3] f32 -> (in-range pts [n,3], src rows [n])."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    n = pts.shape[0]
    out = np.empty((max(n, 1), 3), dtype=np.float32)
    src = np.empty((max(n, 1),), dtype=np.int32)
    lim = np.ascontiguousarray(lim, dtype=np.float32)
    m = lib().dcf_oracle_range_filter(_fp(pts), n, _fp(lim), _fp(out), _ip(src))
    return out[:m].copy(), src[:m].copy()


def voxelize(pts_in, aff, dims, mode="compat", want_ids=False):
    """data_import_carla.py:236-258.  pts_in = in-range points [n,3]; grid [Cz,L,W] f32."""
    pts_in = np.ascontiguousarray(pts_in, dtype=np.float32)
    n = pts_in.shape[0]
    Cz, L, W = dims
    grid = np.empty((Cz, L, W), dtype=np.float32)
    ids = np.empty((3, max(n, 1)), dtype=np.int64) if want_ids else None
    aff = np.ascontiguousarray(aff, dtype=np.float32)
    lib().dcf_oracle_voxelize(_fp(pts_in), n, _fp(aff), Cz, L, W, {"compat": 0, "accum": 1, "occupancy": 2}[mode], _fp(grid),
                              ids.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)) if want_ids else None)
    if want_ids:
        return grid, ids[:, :n].copy()
    return grid


def project(pts_in, crt, ulim, vlim, mode="compat"):
    """data_import_carla.py:196-210.  crt = CRT_tensor [4,3]; returns (uv [m,2], xyz [m,3], src [m])."""
    pts_in = np.ascontiguousarray(pts_in, dtype=np.float32)
    n = pts_in.shape[0]
    crt = np.ascontiguousarray(crt, dtype=np.float32).reshape(12)
    uv = np.empty((max(n, 1), 2), dtype=np.float32)
    xyz = np.empty((max(n, 1), 3), dtype=np.float32)
    src = np.empty((max(n, 1),), dtype=np.int32)
    m = lib().dcf_oracle_project(_fp(pts_in), n, _fp(crt), float(ulim), float(vlim),
                                 0 if mode == "compat" else 1, _fp(uv), _fp(xyz), _ip(src))
    return uv[:m].copy(), xyz[:m].copy(), src[:m].copy()


def knn_bev(xyz, K, h, w, stride, aff, rmax=None):
    """SURVEY.md App. D brute-force KNN.  xyz [n,3] valid points -> int32 [K,h,w]."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    out = np.empty((K, h, w), dtype=np.int32)
    r2 = -1.0 if rmax is None else float(np.float32(rmax) * np.float32(rmax))
    lib().dcf_oracle_knn_bev(_fp(xyz) if n else _fp(np.zeros((1, 3), np.float32)), n, K, h, w, stride,
                             float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]), r2, _ip(out))
    return out


def knn_pixels(xyz, K, pi, pj, stride, aff, rmax=None):
    """Brute-force KNN (same contract as knn_bev) of the listed pixels (pi[q], pj[q]) only -> int32 [npix, K]."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    pi = np.ascontiguousarray(pi, dtype=np.int32)
    pj = np.ascontiguousarray(pj, dtype=np.int32)
    n = xyz.shape[0]
    out = np.empty((pi.shape[0], K), dtype=np.int32)
    r2 = -1.0 if rmax is None else float(np.float32(rmax) * np.float32(rmax))
    lib().dcf_oracle_knn_pixels(_fp(xyz) if n else _fp(np.zeros((1, 3), np.float32)), n, K, _ip(pi), _ip(pj), pi.shape[0], stride,
                                float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]), r2, _ip(out))
    return out


def voxelization_projection(pts, cfg, crt, max_num_pc=None, voxel_mode="compat", proj_mode="compat"):
    """Whole data_import_carla.py:212-267 in one call (the Dataset-side contract).

    Returns (voxel [Cz,L,W], pointcloud_raw [max_num_pc,3], uv [max_num_pc,2], n_valid, ids [3,n_in]).
    """
    g = grid_constants(cfg)
    pin, _ = range_filter(pts, g["lim"])
    grid, ids = voxelize(pin, g["aff"], g["dims"], voxel_mode, want_ids=True)
    if proj_mode == "compat":   # (sic) u against image_height, v against image_width: :202-205
        ulim, vlim = cfg["image_height"], cfg["image_width"]
    else:                       # "correct": u < W, v < H, depth > 0
        ulim, vlim = cfg["image_width"], cfg["image_height"]
    uv, xyz, _ = project(pin, crt, ulim, vlim, proj_mode)
    mp = cfg["max_num_pc"] if max_num_pc is None else max_num_pc
    n = uv.shape[0]
    if n > mp:
        raise RuntimeError("more than max_num_pc points survive (data_import_carla.py:263-266)")
    pc = np.zeros((mp, 3), np.float32); pc[:n] = xyz
    uvp = np.zeros((mp, 2), np.float32); uvp[:n] = uv
    return grid, pc, uvp, n, ids
