"""KittiDataset: the dataset class the reference's train.py names but never wrote (train.py:12, :61-63).

Reads the KITTI 3-D object layout

    <root>/training|testing/velodyne/%06d.bin   float32 [N,4] (x, y, z, reflectance) in the velodyne frame
                           /image_2/%06d.png    left colour camera
                           /calib/%06d.txt      P2 (3x4), R0_rect (3x3), Tr_velo_to_cam (3x4)
                           /label_2/%06d.txt    one object per line (training only)

and returns samples of the CarlaDataset contract (data_import_carla.py:67-82 of the reference): image uint8
[3,H,W] in BGR (the reference decodes with cv2, :296), bboxes [max_num_bbox,9] = (x, y, z, l, w, h, yaw, class, 1)
in the LiDAR frame with the reference's car class id 6 (data_import_carla.py:152), num_bboxes.  With raw=True
(the FrameLoader path) the sample carries the raw points and the frame's own projection
matrix instead of a voxel grid: points and pixels, not 72-MB grids, cross PCIe, and the geometry runs on the GPU.

Host-side numpy only; nothing here is on the device hot path.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

CAR_CLASS = 6                         # data_import_carla.py:152 keeps only class 6
KITTI_CAR_TYPES = ("Car", "Van")


def read_calib(path):
    """{name: float64 array} of a KITTI calib file ('P2: 7.2e+02 0 ...')."""
    out = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            k, v = line.split(":", 1)
            v = v.split()
            if v:
                out[k.strip()] = np.array([float(t) for t in v], dtype=np.float64)
    return out


def velo_to_rect(calib):
    """4x4 velodyne -> rectified camera 0 transform, R0_rect . Tr_velo_to_cam."""
    tr = np.eye(4)
    tr[:3, :4] = calib["Tr_velo_to_cam"].reshape(3, 4)
    r0 = np.eye(4)
    r0[:3, :3] = calib["R0_rect"].reshape(3, 3)
    return r0 @ tr


def crt_from_calib(calib):
    """[4,3] float32 matrix in the layout of CarlaDataset.CRT_tensor (data_import_carla.py:31-34):
    [x, y, z, 1] . CRT = [u*d, v*d, d] with the full P2 . R0_rect . Tr_velo_to_cam chain (translations included)."""
    p2 = calib["P2"].reshape(3, 4)
    return np.ascontiguousarray((p2 @ velo_to_rect(calib)).T.astype(np.float32))


def wrap_yaw(a):
    """data_import_carla.py:138-143: yaw folded into [0, 3.141592]."""
    while a > 3.141592:
        a -= 3.141592
    while a < 0:
        a += 3.141592
    return a


def read_labels(path, calib, config, types=KITTI_CAR_TYPES):
    """label_2 file -> (boxes [max_num_bbox,9] f32, n).  KITTI gives (h, w, l), the bottom-centre location in the
    rectified camera frame and rotation_y about the camera's y axis; the boxes come back in the velodyne frame with
    the centre at mid-height, yaw = -ry - pi/2 folded like the reference folds CARLA yaws."""
    mx = int(config["max_num_bbox"])
    out = torch.zeros(mx, 9)
    n = 0
    if not os.path.isfile(path):
        return out, 0
    inv = np.linalg.inv(velo_to_rect(calib))
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) < 15 or t[0] not in types or n >= mx:
                continue
            h, w, l = float(t[8]), float(t[9]), float(t[10])
            loc = np.array([float(t[11]), float(t[12]) - h / 2.0, float(t[13]), 1.0])
            x, y, z = (inv @ loc)[:3]
            if not (config["lidar_x_min"] <= x < config["lidar_x_max"] and config["lidar_y_min"] <= y < config["lidar_y_max"]):
                continue                                              # valid_bbox, data_import_carla.py:106-110
            yaw = wrap_yaw(-float(t[14]) - np.pi / 2.0)
            out[n] = torch.tensor([x, y, z, l, w, h, yaw, CAR_CLASS, 1], dtype=torch.float32)
            n += 1
    return out, n


def fit_image(rgb, H, W):
    """uint8 [h,w,3] RGB -> [3,H,W] BGR, cropped / zero-padded at the bottom-right so pixel (u,v) keeps its
    calibration (KITTI frames vary between 1224x370 and 1242x376)."""
    out = np.zeros((H, W, 3), dtype=np.uint8)
    h, w = min(H, rgb.shape[0]), min(W, rgb.shape[1])
    out[:h, :w] = rgb[:h, :w, ::-1]
    return torch.from_numpy(np.ascontiguousarray(out.transpose(2, 0, 1)))


class KittiDataset(Dataset):
    def __init__(self, config=None, mode="train", root=None, raw=False, want_bev_image=False):
        super(KittiDataset, self).__init__()
        if mode not in ("train", "test"):
            raise ValueError("mode must be 'train' or 'test'")
        if config is None:                                      # train.py:62 constructs it without arguments
            import yaml
            here = os.path.dirname(os.path.abspath(__file__))
            with open(os.path.join(here, "config", "config_carla.yaml")) as f:
                config = yaml.safe_load(f)
        self.config, self.mode, self.raw, self.want_bev_image = config, mode, bool(raw), bool(want_bev_image)
        root = root or config.get("kitti_root") or config["train_data_dir"]
        self.split_dir = os.path.join(root, "training" if mode == "train" else "testing")
        velo = os.path.join(self.split_dir, "velodyne")
        self.ids = sorted(f[:-4] for f in os.listdir(velo) if f.endswith(".bin")) if os.path.isdir(velo) else []
        self._geometry = None

    def __len__(self):
        return len(self.ids)

    @property
    def geometry(self):
        """Device-side voxeliser / projector (needs the HIP library; created on first use so that raw-mode worker
        processes never touch the GPU)."""
        if self._geometry is None:
            from .data_import_carla import FrameGeometry
            self._geometry = FrameGeometry(self.config)
        return self._geometry

    def read_frame(self, idx):
        """(points [N,3] f32, image [3,H,W] u8, boxes, n, crt [4,3] f32) -- host tensors."""
        from PIL import Image
        fid = self.ids[idx]
        pts = np.fromfile(os.path.join(self.split_dir, "velodyne", fid + ".bin"), dtype=np.float32).reshape(-1, 4)[:, :3]
        rgb = np.asarray(Image.open(os.path.join(self.split_dir, "image_2", fid + ".png")).convert("RGB"))
        calib = read_calib(os.path.join(self.split_dir, "calib", fid + ".txt"))
        boxes, n = read_labels(os.path.join(self.split_dir, "label_2", fid + ".txt"), calib, self.config)
        image = fit_image(rgb, int(self.config["image_height"]), int(self.config["image_width"]))
        return torch.from_numpy(np.ascontiguousarray(pts)), image, boxes, n, torch.from_numpy(crt_from_calib(calib))

    def __getitem__(self, idx):
        if idx < 0 or idx >= len(self.ids):
            raise IndexError("idx is not in data file")
        pts, image, boxes, n, crt = self.read_frame(idx)
        if self.raw:
            return {"image": image, "bboxes": boxes, "num_bboxes": n, "lidar_points": pts, "crt": crt}
        g = self.geometry
        voxel = g.voxelize(pts)
        pc, uv, cnt = g.project(pts, crt=crt.numpy())
        sample = {"image": image, "bboxes": boxes, "num_bboxes": n, "pointcloud_raw": pc, "projected_loc_uv": uv,
                  "num_points_raw": cnt, "pointcloud": voxel}
        if self.want_bev_image:
            from . import ops
            pin, _, _ = ops.range_filter(pts.cuda(), g.grid.lim)
            img = torch.zeros(3, self.config["voxel_length"], self.config["voxel_width"], device=pin.device)
            if pin.numel():
                img[:, (pin[:, 0] * g.grid.xs + g.grid.xo).long(), (pin[:, 1] * g.grid.ys + g.grid.yo).long()] = 1
            sample["lidar_bev_2Dimage"] = img
        return sample
