"""Data path: KittiDataset, CarlaDataset raw mode over (stubbed) h5py, FrameLoader.

Host logic runs without a GPU; the last test drives the loader into a train step on the device."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from _util import golden_cfg, load_golden, pkg


# ------------------------------------------------------------------ a tiny KITTI tree written by the test
def kitti_calib(seed=0):
    rs = np.random.RandomState(seed)
    P2 = np.array([[60.0, 0.0, 64.0, 4.5 + seed], [0.0, 60.0, 48.0, 0.2], [0.0, 0.0, 1.0, 0.003]])
    a = 0.01 * (seed + 1)
    R0 = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
    Tr = np.concatenate([pkg("calib").R_LIDAR_TO_CAM, rs.uniform(-0.3, 0.3, (3, 1))], 1)
    return P2, R0, Tr


def write_kitti(root, n_frames, cfg, n_points=700, image_hw=(94, 126)):
    det = pkg("detfill")
    d = os.path.join(root, "training")
    for sub in ("velodyne", "image_2", "calib", "label_2"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    from PIL import Image
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    truth = []
    for i in range(n_frames):
        fid = "%06d" % i
        pts = det.synthetic_points(n_points + 37 * i, lim6, 300 + i)
        np.concatenate([pts, np.full((pts.shape[0], 1), 0.5, np.float32)], 1).astype(np.float32).tofile(os.path.join(d, "velodyne", fid + ".bin"))
        rgb = np.random.RandomState(i).randint(0, 256, image_hw + (3,)).astype(np.uint8)
        Image.fromarray(rgb).save(os.path.join(d, "image_2", fid + ".png"))
        P2, R0, Tr = kitti_calib(i)
        with open(os.path.join(d, "calib", fid + ".txt"), "w") as f:
            f.write("P0: " + " ".join(["0"] * 12) + "\n")
            f.write("P2: " + " ".join("%.12e" % v for v in P2.reshape(-1)) + "\n")
            f.write("R0_rect: " + " ".join("%.12e" % v for v in R0.reshape(-1)) + "\n")
            f.write("Tr_velo_to_cam: " + " ".join("%.12e" % v for v in Tr.reshape(-1)) + "\n")
        # two cars given in the LiDAR frame, written as KITTI camera-frame labels by the forward transform
        boxes = [(0.5 * (lim6[0] + lim6[1]) + i, 0.25 * lim6[3], -1.0, 4.2, 1.8, 1.5, 0.3 + 0.2 * i),
                 (0.3 * lim6[1], 0.5 * lim6[2], -0.8, 3.9, 1.7, 1.6, 2.1)]
        M = np.eye(4); M[:3, :3] = R0
        T4 = np.eye(4); T4[:3, :4] = Tr
        with open(os.path.join(d, "label_2", fid + ".txt"), "w") as f:
            f.write("Pedestrian 0 0 0 1 2 3 4 1.7 0.5 0.5 1 1 10 0\n")
            for (x, y, z, l, w, h, yaw) in boxes:
                c = M @ T4 @ np.array([x, y, z, 1.0])
                ry = -yaw - np.pi / 2
                f.write("Car 0.0 0 0.0 0 0 10 10 %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % (h, w, l, c[0], c[1] + h / 2, c[2], ry))
            f.write("Car 0.0 0 0.0 0 0 10 10 1.5 1.8 4.0 0.0 1.0 %.3f 0.0\n" % (lim6[1] + 50.0))     # outside the grid: dropped
        truth.append(dict(pts=pts, rgb=rgb, boxes=boxes, calib=(P2, R0, Tr)))
    return truth


def tiny_cfg():
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    cfg.update(dict(image_height=96, image_width=128, max_num_pc=2048, projection_mode="correct", dataset_name="kitti"))
    return cfg


def test_kitti_dataset_raw_sample(tmp_path):
    cfg = tiny_cfg()
    truth = write_kitti(str(tmp_path), 3, cfg)
    K = pkg("data_import_kitti")
    ds = K.KittiDataset(cfg, root=str(tmp_path), raw=True)
    assert len(ds) == 3
    for i in range(3):
        s = ds[i]
        t = truth[i]
        assert set(s) == {"image", "bboxes", "num_bboxes", "lidar_points", "crt"}
        assert s["lidar_points"].dtype == torch.float32 and np.array_equal(s["lidar_points"].numpy(), t["pts"])
        # image: BGR, [3,H,W] u8, zero padded to the configured size with pixel (u,v) in place
        im = s["image"].numpy()
        assert im.shape == (3, 96, 128) and im.dtype == np.uint8
        assert np.array_equal(im[:, :94, :126], t["rgb"][:, :, ::-1].transpose(2, 0, 1))
        assert not im[:, 94:, :].any() and not im[:, :, 126:].any()
        # boxes come back in the LiDAR frame (the far one and the pedestrian are dropped)
        assert s["num_bboxes"] == 2
        got = s["bboxes"][:2].numpy()
        for g, (x, y, z, l, w, h, yaw) in zip(got, t["boxes"]):
            assert np.allclose(g[:6], [x, y, z, l, w, h], atol=1e-4)
            assert abs(g[6] - yaw) < 1e-4 and g[7] == 6 and g[8] == 1
        assert not s["bboxes"][2:].any()
        # projection matrix: [x,y,z,1] . CRT equals the P2 . R0 . Tr chain
        P2, R0, Tr = t["calib"]
        p = np.array([12.0, -3.0, 0.4, 1.0])
        M = np.eye(4); M[:3, :3] = R0
        T4 = np.eye(4); T4[:3, :4] = Tr
        assert np.allclose(p @ s["crt"].numpy().astype(np.float64), P2 @ M @ T4 @ p, rtol=1e-5, atol=1e-4)
    with pytest.raises(IndexError):
        ds[3]
    assert len(K.KittiDataset(cfg, mode="test", root=str(tmp_path), raw=True)) == 0


def test_frame_loader_host_batches_and_sharding(tmp_path):
    cfg = tiny_cfg()
    truth = write_kitti(str(tmp_path), 7, cfg)
    ds = pkg("data_import_kitti").KittiDataset(cfg, root=str(tmp_path), raw=True)
    FL = pkg("frame_loader")
    loader = FL.FrameLoader(ds, 2, num_workers=2, device=None)
    assert len(loader) == 4
    seen = 0
    for batch in loader:
        B = len(batch["points"])
        assert batch["image"].shape == (B, 3, 96, 128) and batch["bboxes"].shape == (B, cfg["max_num_bbox"], 9)
        assert batch["num_bboxes"].tolist() == [2] * B and len(batch["crt"]) == B
        for p in batch["points"]:                        # ragged point lists survive the collate
            assert np.array_equal(p.numpy(), truth[seen]["pts"])
            seen += 1
    assert seen == 7
    # DistributedSampler shards the frames across ranks: disjoint, complete (padded by wrap-around)
    got = []
    for rank in range(2):
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=False)
        ids = []
        for batch in FL.FrameLoader(ds, 2, sampler=sampler, device=None):
            ids += [int(p.shape[0]) for p in batch["points"]]
        got.append(ids)
    sizes = [t["pts"].shape[0] for t in truth]
    assert len(got[0]) == len(got[1]) == 4 and set(got[0] + got[1]) == set(sizes)
    with pytest.raises(ValueError):
        FL.FrameLoader(pkg("data_import_kitti").KittiDataset(cfg, root=str(tmp_path)), 2, device=None)


# ------------------------------------------------------------------ CarlaDataset over a stand-in for h5py
class _FakeH5File(dict):
    """Mapping frame id -> {dataset name -> array}, loaded from a pickle: what data_import_carla.py:163-171 reads."""
    opened = []

    def __init__(self, path, mode="r"):
        with open(path, "rb") as f:
            super(_FakeH5File, self).__init__(pickle.load(f))
        _FakeH5File.opened.append(os.getpid())


def test_carla_dataset_raw_over_hdf5_layout(tmp_path, monkeypatch):
    cfg = tiny_cfg()
    cfg.update(train_data_dir=str(tmp_path), dataset_name="carla")
    rs = np.random.RandomState(3)
    frames = {}
    for i in range(5):
        lidar = np.concatenate([rs.uniform(-1, 1, (40 + i, 3)), rs.uniform(0, 20, (40 + i, 3))], 1)     # columns 3:6 are x, y, z
        obj = np.zeros((3, 10))
        obj[0] = [10.0, 1.0, -1.0, 0, 0, 0.4, 1.8, 4.1, 1.5, 6]      # a car in range
        obj[1] = [10.0, 1.0, -1.0, 0, 0, 0.4, 1.8, 4.1, 1.5, 3]      # not a car
        obj[2] = [-5.0, 1.0, -1.0, 0, 0, 0.4, 1.8, 4.1, 1.5, 6]      # behind the sensor
        frames["%04d" % i] = {"object_data": obj, "lidar_data": lidar,
                               "center_image_data": rs.randint(0, 256, (96, 128, 3)).astype(np.uint8)}
    with open(os.path.join(str(tmp_path), "scenario0.hdf5"), "wb") as f:
        pickle.dump(frames, f)
    fake = types.ModuleType("h5py")
    fake.File = _FakeH5File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    D = pkg("data_import_carla")
    ds = D.CarlaDataset(cfg, raw=True)
    assert len(ds) == 5 and ds._geometry is None                   # raw mode never builds device state
    s = ds[2]
    key = sorted(frames)[2]
    assert np.allclose(s["lidar_points"].numpy(), frames[key]["lidar_data"][:, 3:6].astype(np.float32))
    assert s["image"].shape == (3, 96, 128) and s["num_bboxes"] == 1
    assert np.allclose(s["bboxes"][0].numpy(), [10.0, 1.0, -1.0, 4.1, 1.8, 1.5, 0.4, 6, 1])
    # a pickled copy (what a DataLoader worker receives) reopens its own handles
    n_open = len(_FakeH5File.opened)
    ds2 = pickle.loads(pickle.dumps(ds))
    assert ds2.hdf5_files == {} and len(_FakeH5File.opened) == n_open
    ds2._pid = -1
    assert np.array_equal(ds2[2]["lidar_points"].numpy(), s["lidar_points"].numpy()) and len(_FakeH5File.opened) == n_open + 1
    batches = list(pkg("frame_loader").FrameLoader(ds, 2, device=None))
    assert [len(b["points"]) for b in batches] == [2, 2, 1] and batches[0]["crt"] is None


def test_carla_dataset_without_h5py_fails_loudly(tmp_path, monkeypatch):
    cfg = tiny_cfg()
    cfg.update(train_data_dir=str(tmp_path))
    open(os.path.join(str(tmp_path), "a.hdf5"), "wb").close()
    monkeypatch.setitem(sys.modules, "h5py", None)
    with pytest.raises(ImportError):
        pkg("data_import_carla").CarlaDataset(cfg, raw=True)


# ------------------------------------------------------------------ device leg
@pytest.mark.gpu
def test_frame_loader_feeds_train_steps(tmp_path):
    """Raw KITTI frames -> worker processes -> pinned staging -> copy stream -> geometry on the GPU -> train step.
    The per-frame projection (own calibration per frame) is checked against the CPU restatement, the voxel grids
    against the direct voxeliser."""
    from oracle import geometry_ref
    cfg = tiny_cfg()
    cfg["dtype"] = "f32"
    cfg["batch_size"] = 2
    cfg["fusion"] = dict(enabled=True, K=3, r_max=None, image_channels=64, image_stream="resnet18", zero_init_last=False)
    truth = write_kitti(str(tmp_path), 5, cfg)
    ds = pkg("data_import_kitti").KittiDataset(cfg, root=str(tmp_path), raw=True)
    loader = pkg("frame_loader").FrameLoader(ds, 2, num_workers=2, drop_last=True)
    trainer = pkg("train").Train(cfg)
    pkg("detfill").fill_state_dict(trainer.model)
    seen, losses = 0, []
    for batch in loader:
        batch.wait()
        assert batch["image"].is_cuda and batch["image"].dtype == torch.uint8
        for b, p in enumerate(batch["points"]):
            t = truth[seen + b]
            assert torch.equal(p.cpu(), torch.from_numpy(t["pts"]))
            pc, uv, cnt = ds.geometry.project(p, crt=batch["crt"][b])
            _, pc_ref, uv_ref, n_ref, _ = geometry_ref.voxelization_projection(t["pts"], cfg, batch["crt"][b].numpy(), proj_mode="correct")
            n = int(cnt.item())
            assert n == n_ref and n > 0
            assert np.array_equal(pc[:n].cpu().numpy(), pc_ref[:n])
            assert np.allclose(uv[:n].cpu().numpy(), uv_ref[:n], rtol=0, atol=2e-3)
        x_lidar, geom = trainer.geometry_async(ds.geometry, batch["points"], crts=batch["crt"], wait_event=batch.event)
        torch.cuda.synchronize()
        for b, p in enumerate(batch["points"]):
            assert torch.equal(x_lidar[b], ds.geometry.voxelize(p))
        trainer.one_step_raw(ds.geometry, batch)
        losses.append(float(trainer.loss_value.item()))
        seen += len(batch["points"])
    assert seen == 4 and all(np.isfinite(l) for l in losses)


class _RawFrames(torch.utils.data.Dataset):
    """In-memory raw-mode dataset: frame i has counts[i] points; item `bad` raises."""
    raw = True

    def __init__(self, counts, hw=(24, 40), bad=None):
        g = torch.Generator().manual_seed(5)
        self.pts = [torch.rand((n, 3), generator=g) for n in counts]
        self.img = [torch.randint(0, 256, (3,) + hw, dtype=torch.uint8, generator=g) for _ in counts]
        self.bad = bad

    def __len__(self):
        return len(self.pts)

    def __getitem__(self, i):
        if i == self.bad:
            raise RuntimeError("frame %d is unreadable" % i)
        return {"image": self.img[i], "bboxes": torch.zeros((20, 9)), "num_bboxes": 0, "lidar_points": self.pts[i], "crt": None}


@pytest.mark.gpu
def test_frame_loader_background_staging_equals_inline_staging():
    """The staging thread (six staging sets, four batches of look-ahead) delivers the same device batches, in order, as staging on the calling
    thread -- also when a later frame is larger than the sets sized from the first one, and with work enqueued on the consumer's
    stream between batches (the `consumed` event is what keeps the copy stream off buffers that are still being read)."""
    FL = pkg("frame_loader")
    counts = [700, 300, 512, 900, 4096, 5000, 120, 6000, 8000, 100, 9000, 4500, 777]
    ds = _RawFrames(counts)
    ref = [(ds.pts[i], ds.img[i]) for i in range(len(counts))]
    for threaded in (True, False):
        seen = 0
        sink = torch.zeros((), device="cuda")
        for batch in FL.FrameLoader(ds, 2, threaded=threaded):
            batch.wait()
            for b, p in enumerate(batch["points"]):
                # a slow reader on the consumer's stream: the next-but-one batch must not overwrite what it still reads
                for _ in range(20):
                    sink = sink + p.sum() * 1e-9
                assert torch.equal(p.cpu(), ref[seen + b][0]), (threaded, seen + b)
                assert torch.equal(batch["image"][b].cpu(), ref[seen + b][1])
            seen += len(batch["points"])
        assert seen == len(counts)
        torch.cuda.synchronize()


@pytest.mark.gpu
def test_frame_loader_thread_stops_on_early_exit_and_reports_dataset_errors():
    import threading
    FL = pkg("frame_loader")
    n0 = threading.active_count()
    it = iter(FL.FrameLoader(_RawFrames([256] * 12), 2))
    next(it); next(it)
    it.close()                                   # a `break` out of the for loop: the generator's finally stops the worker
    assert threading.active_count() <= n0
    with pytest.raises(RuntimeError) as e:
        for _ in FL.FrameLoader(_RawFrames([256] * 8, bad=5), 2):
            pass
    assert "unreadable" in str(e.value)
    assert threading.active_count() <= n0
