"""Drop-in counterpart of the reference's data_import_carla.py.

CarlaDataset(config, mode, want_bev_image)[i] returns the same dict keys / shapes / dtypes
(data_import_carla.py:67-82).  The per-frame geometry -- range filter, trilinear voxeliser,
pinhole projection + compaction (data_import_carla.py:196-267), which the reference runs on
the CPU inside the DataLoader -- runs here as HIP kernels on the GPU (ops.voxelize /
ops.project_filter), bit-exact with the reference under deterministic algorithms.

HDF5 scenarios (data_import_carla.py:84-104, :163-171) are read through h5py when it is installed (it is not in
the build image: constructing a CarlaDataset over a directory that holds .hdf5 files raises ImportError there);
file handles are opened per process, so the dataset can sit behind DataLoader workers.  raw=True returns the raw
point list instead of a voxel grid -- the FrameLoader path (frame_loader.py).
SyntheticDataset produces frames of the same contract for benchmarks and tests.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _hip as H
from . import calib, detfill, ops


class FrameGeometry(object):
    """Device-side Voxelization_Projection (data_import_carla.py:212-267) for one config."""

    def __init__(self, config, crt=None):
        self.config = config
        self.grid = ops.GridSpec(config)
        self.crt = calib.carla_crt() if crt is None else np.ascontiguousarray(crt, dtype=np.float32)
        modes = {"compat": H.VOXEL_COMPAT, "accum": H.VOXEL_ACCUM, "occupancy": H.VOXEL_OCCUPANCY}
        if config.get("voxel_mode", "compat") not in modes:
            raise ValueError("voxel_mode must be one of %s" % sorted(modes))
        self.voxel_mode = modes[config.get("voxel_mode", "compat")]
        self.proj_mode = H.PROJ_COMPAT if config.get("projection_mode", "compat") == "compat" else H.PROJ_CORRECT
        self._owner = None

    def limits(self):
        c = self.config
        if self.proj_mode == H.PROJ_COMPAT:          # (sic) data_import_carla.py:202-205
            return c["image_height"], c["image_width"]
        return c["image_width"], c["image_height"]

    def _pts(self, lidar_points):
        return lidar_points.to(device="cuda", dtype=torch.float32).contiguous()

    def voxelize(self, lidar_points, voxel_out=None, mode=None):
        """Voxel grid [Cz,L,W] of one frame (data_import_carla.py:231-258), optionally written into voxel_out.
        mode: override of the configured voxel mode (H.VOXEL_OCCUPANCY = the reference's interpolate=False)."""
        pts = self._pts(lidar_points)
        g = self.grid
        owner = None
        mode = self.voxel_mode if mode is None else mode
        if mode == H.VOXEL_COMPAT:
            # one owner-map workspace per (device, stream): frames voxelised on different streams must not share it
            key = (pts.device, H.stream_ptr())
            if self._owner is None:
                self._owner = {}
            owner = self._owner.get(key)
            if owner is None:
                owner = self._owner[key] = torch.zeros((2, g.dims[0] * g.dims[1] * g.dims[2]), dtype=torch.int32, device=pts.device)
        return ops.voxelize(pts, g.lim, g.aff, g.dims, mode, owner, voxel_out)

    def voxelize_batch(self, points_list, out, nhwc_dtype=None):
        """Voxel grids of the frames of a batch written into out [B,Cz,L,W] fp32 -- or, with nhwc_dtype (a dtype code), into
        out [B,L,W,Cz] of that type: the engine's input image, without the fp32 grid and its transpose (compat mode, <= 8
        frames).  Compat mode runs all frames in one launch per phase, other modes frame by frame."""
        if nhwc_dtype is not None:
            if self.voxel_mode != H.VOXEL_COMPAT or len(points_list) > 8:
                raise H.DcfError("the NHWC voxel image needs compat mode and at most 8 frames per call")
        elif self.voxel_mode != H.VOXEL_COMPAT or len(points_list) > 8:
            for b, p in enumerate(points_list):
                self.voxelize(p, out[b])
            return out
        pts = [self._pts(p) for p in points_list]
        g = self.grid
        key = (pts[0].device, H.stream_ptr(), len(pts))
        if self._owner is None:
            self._owner = {}
        owner = self._owner.get(key)
        if owner is None:
            owner = self._owner[key] = torch.zeros((len(pts), 2, g.dims[0] * g.dims[1] * g.dims[2]), dtype=torch.int32, device=pts[0].device)
        if nhwc_dtype is not None:
            return ops.voxelize_batch_nhwc(nhwc_dtype, pts, g.lim, g.aff, g.dims, owner, out)
        return ops.voxelize_batch(pts, g.lim, g.aff, g.dims, owner, out)

    def project(self, lidar_points, crt=None, out=None):
        """(pointcloud_raw [max_num_pc,3], uv [max_num_pc,2], n_valid int32[1] on device) of one frame
        (data_import_carla.py:196-210, :262-266).  crt: this frame's own [4,3] matrix (KITTI calibrates per frame).
        out: (xyz [max_num_pc,3], uv [max_num_pc,2], count [1]) zero-filled slices of a batch tensor to write into (used
        when the frame has at most max_num_pc points)."""
        pts = self._pts(lidar_points)
        ulim, vlim = self.limits()
        mp_ = int(self.config["max_num_pc"])
        crt = self.crt if crt is None else np.ascontiguousarray(crt, dtype=np.float32)
        if out is not None and pts.shape[0] <= mp_:
            ops.project_filter(pts, self.grid.lim, crt, ulim, vlim, self.proj_mode, n_out=mp_, out=(out[1], out[0], out[2]))
            return out
        n_out = max(mp_, pts.shape[0])
        uv, xyz, cnt, _ = ops.project_filter(pts, self.grid.lim, crt, ulim, vlim, self.proj_mode, n_out=n_out)
        mp = int(self.config["max_num_pc"])
        return xyz[:mp], uv[:mp], cnt

    def project_batch(self, points_list, crts, xyz_all, uv_all, cnt_all):
        """project() for the frames of a batch in one launch per phase (dcf_project_filter_batch): frames with at most max_num_pc
        points, <= 8 of them; xyz_all [B,max_num_pc,3], uv_all [B,max_num_pc,2] zero-filled, cnt_all int32 [B].  Returns False when
        the batch does not qualify (the caller then projects frame by frame)."""
        mp_ = int(self.config["max_num_pc"])
        pts = [self._pts(p) for p in points_list]
        if not (1 <= len(pts) <= 8) or any(p.shape[0] > mp_ for p in pts):
            return False
        ulim, vlim = self.limits()
        mats = np.stack([np.ascontiguousarray(self.crt if (crts is None or crts[b] is None) else crts[b], dtype=np.float32).reshape(12)
                         for b in range(len(pts))], 0)
        self._proj_ws = ops.project_filter_batch(pts, self.grid.lim, mats, ulim, vlim, self.proj_mode, uv_all, xyz_all, cnt_all,
                                                 getattr(self, "_proj_ws", None))
        return True

    def __call__(self, lidar_points, want_ids=False, voxel_out=None, voxel_mode=None):
        """lidar_points [N,3] f32 (any device) -> (voxel [Cz,L,W], pointcloud_raw [max_num_pc,3],
        uv [max_num_pc,2], n_valid int32[1] on device, ids or None).  voxel_out: optional [Cz,L,W] slice of a
        batch tensor to write the grid into (saves the stack copy)."""
        pts = self._pts(lidar_points)
        voxel = self.voxelize(pts, voxel_out, voxel_mode)
        xyz, uv, cnt = self.project(pts)
        ids = None
        if want_ids:
            pin, _, c2 = ops.range_filter(pts, self.grid.lim)
            ids = pin  # in-range points; trunc'd voxel ids are derived on demand by getLidarImage
        return voxel, xyz, uv, cnt, ids


class CarlaDataset(Dataset):
    def __init__(self, config, mode="train", want_bev_image=False, raw=False):
        super(CarlaDataset, self).__init__()
        self.config = config
        self.mode = mode
        self.raw = bool(raw)
        self.want_bev_image = bool(want_bev_image)
        self._geometry = None
        self.CRT_tensor = torch.from_numpy(calib.carla_crt())
        self._pid = None
        self.hdf5_files = self.load_dataset(mode)
        self._pid = os.getpid()
        self.hdf5_id_dict = dict((k, list(v.keys())) for k, v in self.hdf5_files.items())
        self.scenario_name = list(self.hdf5_files.keys())
        self.scenario_length = [len(self.hdf5_files[k]) for k in self.scenario_name]
        self.length = sum(self.scenario_length)

    def __len__(self):
        return self.length

    @property
    def geometry(self):
        """Device-side voxeliser / projector, created on first use: raw-mode worker processes never touch the GPU."""
        if self._geometry is None:
            self._geometry = FrameGeometry(self.config)
        return self._geometry

    def __getstate__(self):
        d = dict(self.__dict__)
        d["hdf5_files"], d["_pid"], d["_geometry"] = {}, None, None      # handles and device state are per process
        return d

    def _files(self):
        if self._pid != os.getpid():                    # forked / spawned worker: its own read-only handles
            self.hdf5_files = self.load_dataset(self.mode)
            self._pid = os.getpid()
        return self.hdf5_files

    def load_dataset(self, mode="train"):
        if mode not in ("train", "test"):
            raise ValueError("mode must be 'train' or 'test'")
        path = self.config["train_data_dir"] if mode == "train" else self.config["test_data_dir"]
        files = {}
        names = [f for f in sorted(os.listdir(path)) if f.split(".")[-1] == "hdf5"] if os.path.isdir(path) else []
        if names:
            try:
                import h5py
            except ImportError as e:
                raise ImportError("reading CARLA .hdf5 scenarios needs h5py (not installed): %s" % e)
            for f in names:
                try:
                    files[f] = h5py.File(os.path.join(path, f), "r")
                except OSError:
                    print(f + " doesnt work. we except this folder")
        return files

    # label packing, data_import_carla.py:106-161 (class 6 = car only, yaw wrapped into (0, 3.141592))
    def valid_bbox(self, o):
        c = self.config
        return c["lidar_x_min"] <= o[0] < c["lidar_x_max"] and c["lidar_y_min"] <= o[1] < c["lidar_y_max"]

    @staticmethod
    def orientation_inner_bound(ori):
        while ori > 3.141592:
            ori -= 3.141592
        while ori < 0:
            ori += 3.141592
        return ori

    def arangeLabelData(self, object_datas):
        out = torch.zeros(self.config["max_num_bbox"], 9)
        i = 0
        for o in object_datas:
            if i >= self.config["max_num_bbox"]:
                break
            if not self.valid_bbox(o) or o[9] != 6:
                continue
            out[i, :] = torch.tensor([o[0], o[1], o[2], o[7], o[6], o[8], self.orientation_inner_bound(float(o[5])), o[9], 1])
            i += 1
        return out, i

    def getOneStepData(self, data, id):
        obj = torch.tensor(np.array(data[id]["object_data"]))
        lidar = torch.tensor(np.array(data[id]["lidar_data"])).type(torch.float)[:, 3:6]
        image = torch.tensor(np.array(data[id]["center_image_data"]))
        return obj, lidar, image

    def Voxelization_Projection(self, lidar_data, interpolate=True):
        """interpolate=False: occupancy grid (the voxel of the trunc'd ids := 1, data_import_carla.py:231-234)."""
        voxel, pc, uv, cnt, ids = self.geometry(lidar_data, want_ids=self.want_bev_image,
                                                voxel_mode=None if interpolate else H.VOXEL_OCCUPANCY)
        return voxel, pc, uv, cnt, ids

    def getLidarImage(self, in_range_points):
        g = self.geometry.grid
        img = torch.zeros(3, self.config["voxel_length"], self.config["voxel_width"], device=in_range_points.device)
        if in_range_points is not None and in_range_points.numel():
            ix = (in_range_points[:, 0] * g.xs + g.xo).long()
            iy = (in_range_points[:, 1] * g.ys + g.yo).long()
            img[:, ix, iy] = 1
        return img

    def __getitem__(self, idx):
        if idx >= self.length or idx < 0:
            raise IndexError("idx is not in data file")
        k = idx
        for name, n in zip(self.scenario_name, self.scenario_length):
            if k >= n:
                k -= n
                continue
            fid = self.hdf5_id_dict[name][k].strip()
            obj, lidar, image = self.getOneStepData(self._files()[name], fid)
            boxes, nb = self.arangeLabelData(obj)
            if self.raw:
                return {"image": image.permute(2, 0, 1).contiguous(), "bboxes": boxes, "num_bboxes": nb,
                        "lidar_points": lidar.contiguous(), "crt": None}
            voxel, pc, uv, cnt, ids = self.Voxelization_Projection(lidar)
            sample = {"image": image.permute(2, 0, 1), "bboxes": boxes, "num_bboxes": nb, "pointcloud_raw": pc,
                      "projected_loc_uv": uv, "num_points_raw": cnt, "pointcloud": voxel}
            if self.want_bev_image:
                sample["lidar_bev_2Dimage"] = self.getLidarImage(ids)
            return sample
        raise IndexError(idx)


def synthetic_boxes(config, seed, n=8):
    """8 car boxes per frame: class 6, centre U(grid), l~U(3.5,4.8), w~U(1.6,2.1), h~U(1.4,1.8), yaw~U(0,pi)."""
    c = config
    u = detfill.uniform((n, 6), 0xB0C5 + int(seed) * 31, 0.0, 1.0)
    out = torch.zeros(c["max_num_bbox"], 9)
    for i in range(min(n, c["max_num_bbox"])):
        x = c["lidar_x_min"] + 2.0 + u[i, 0] * (c["lidar_x_max"] - c["lidar_x_min"] - 4.0)
        y = c["lidar_y_min"] + 2.0 + u[i, 1] * (c["lidar_y_max"] - c["lidar_y_min"] - 4.0)
        out[i] = torch.tensor([x, y, -1.0, 3.5 + 1.3 * u[i, 2], 1.6 + 0.5 * u[i, 3], 1.4 + 0.4 * u[i, 4], 3.14159 * u[i, 5], 6, 1])
    return out, min(n, c["max_num_bbox"])


class SyntheticDataset(Dataset):
    """Frames of the CarlaDataset contract from the deterministic generator of SURVEY.md 8(d)."""

    def __init__(self, config, length=16, num_points=None, crt=None, image_hw=None, raw=False):
        self.config, self.length, self.raw = config, length, bool(raw)
        self.num_points = num_points or config["max_num_pc"]
        self._crt, self._geometry = crt, None
        self.image_hw = image_hw or (config["image_height"], config["image_width"])
        c = config
        self.lim6 = (c["lidar_x_min"], c["lidar_x_max"], c["lidar_y_min"], c["lidar_y_max"], c["lidar_z_min"], c["lidar_z_max"])

    def __len__(self):
        return self.length

    @property
    def geometry(self):
        if self._geometry is None:
            self._geometry = FrameGeometry(self.config, self._crt)
        return self._geometry

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_geometry"] = None
        return d

    def raw_frame(self, idx):
        pts = torch.from_numpy(detfill.synthetic_points(self.num_points, self.lim6, 1234 + idx))
        img = torch.from_numpy(detfill.synthetic_image(self.image_hw[0], self.image_hw[1], 1234 + idx))
        boxes, nb = synthetic_boxes(self.config, 1234 + idx)
        return pts, img, boxes, nb

    def __getitem__(self, idx):
        pts, img, boxes, nb = self.raw_frame(idx)
        if self.raw:
            return {"image": img, "bboxes": boxes, "num_bboxes": nb, "lidar_points": pts, "crt": None}
        voxel, pc, uv, cnt, _ = self.geometry(pts)
        return {"image": img, "bboxes": boxes, "num_bboxes": nb, "pointcloud_raw": pc, "projected_loc_uv": uv,
                "num_points_raw": cnt, "pointcloud": voxel}
