"""CPU suite: independent pins for the parts the reference lacks (reference model.py:192 names
torchvision's resnet18, model.py:199-203 is the fusion TODO).  No reference code exists for them, so the oracle's statements
are checked here against PUBLISHED third-party operators that torch / scipy ship on the CPU:

  bilinear gather   oracle/model_ref.bilinear_sample   ==  torch.nn.functional.grid_sample(align_corners=False, padding_mode="border")
  camera trunk      oracle/model_ref.image_stream      ==  a literal torch.nn build of the published ResNet-18 / ResNet-50 topology
                                                           (conv / BatchNorm2d / MaxPool2d modules, torchvision's key names and
                                                           parameter counts) loaded with the same state dict
  KNN               oracle/dcf_oracle.c brute force    ==  scipy.spatial.cKDTree.query (float64) at cfg2 size

The product's HIP kernels are checked against these oracle functions in the GPU suite; this file is what ties the oracle itself
to something other than this repository's own statement."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import geometry_ref, model_ref  # noqa: E402


# ------------------------------------------------------------------------------------------------ bilinear gather
@pytest.mark.parametrize("shape", [(64, 94, 311), (7, 5, 9), (3, 1, 4)])
def test_bilinear_sample_equals_grid_sample_border(shape):
    C, Hf, Wf = shape
    g = torch.Generator().manual_seed(5)
    fmap = torch.randn(C, Hf, Wf, generator=g)
    n = 4000
    # image-pixel positions over the stride-4 map's whole extent and beyond it on every side, plus the exact corners / edges
    u = (torch.rand(n, generator=g) * (Wf + 2.0) - 1.0) * 4.0
    v = (torch.rand(n, generator=g) * (Hf + 2.0) - 1.0) * 4.0
    edge = torch.tensor([[0.0, 0.0], [2.0, 2.0], [4.0 * Wf, 4.0 * Hf], [4.0 * Wf - 2.0, 4.0 * Hf - 2.0], [-3.0, 4.0 * Hf + 9.0], [2.0, 4.0 * Hf - 2.0]])
    uv = torch.cat([torch.stack([u, v], 1), edge], 0)
    mine = model_ref.bilinear_sample(fmap, uv)                                  # [n, C]
    # grid_sample's normalised coordinates under align_corners=False: index-space x = ((g + 1) * W - 1) / 2, i.e. g = 2 (x + 0.5) / W - 1;
    # the oracle's index-space position is u / 4 - 0.5  =>  g = 2 (u / 4) / W - 1
    # (evaluated in float64: in fp32 the round trip through [-1, 1] alone moves a position by ~1e-5 pixels at Wf = 311)
    uvd = uv.double()
    gx = 2.0 * (uvd[:, 0] * 0.25) / Wf - 1.0
    gy = 2.0 * (uvd[:, 1] * 0.25) / Hf - 1.0
    grid = torch.stack([gx, gy], 1).view(1, -1, 1, 2)
    ref = F.grid_sample(fmap.double().unsqueeze(0), grid, mode="bilinear", padding_mode="border", align_corners=False)[0, :, :, 0].t()
    err = float((mine.double() - ref).abs().max())
    assert err < 2e-6, "bilinear_sample vs grid_sample(border, align_corners=False): max abs %g" % err


# ------------------------------------------------------------------------------------------------ camera trunk
class _Basic(nn.Module):
    """BasicBlock of He et al. 2016 as torchvision publishes it: 3x3(stride) - BN - ReLU - 3x3 - BN, + shortcut, ReLU."""
    expansion = 1

    def __init__(self, cin, w, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, w, 3, stride, 1, bias=False); self.bn1 = nn.BatchNorm2d(w)
        self.conv2 = nn.Conv2d(w, w, 3, 1, 1, bias=False); self.bn2 = nn.BatchNorm2d(w)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = down

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _Bottleneck(nn.Module):
    """Bottleneck, torchvision's "v1.5": 1x1 - 3x3(stride) - 1x1 (x4), the stride on the 3x3."""
    expansion = 4

    def __init__(self, cin, w, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, w, 1, bias=False); self.bn1 = nn.BatchNorm2d(w)
        self.conv2 = nn.Conv2d(w, w, 3, stride, 1, bias=False); self.bn2 = nn.BatchNorm2d(w)
        self.conv3 = nn.Conv2d(w, 4 * w, 1, bias=False); self.bn3 = nn.BatchNorm2d(4 * w)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = down

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class _Trunk(nn.Module):
    """The published ResNet trunk (7x7 / 2 stem, 3x3 / 2 max-pool, four stages; no avg-pool / fc), module names as torchvision's."""

    def __init__(self, block, layers):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.cin = 64
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)

    def _make(self, block, w, n, stride):
        down = None
        if stride != 1 or self.cin != w * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.cin, w * block.expansion, 1, stride, bias=False), nn.BatchNorm2d(w * block.expansion))
        mods = [block(self.cin, w, stride, down)]
        self.cin = w * block.expansion
        for _ in range(1, n):
            mods.append(block(self.cin, w, 1, None))
        return nn.Sequential(*mods)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        c2 = self.layer1(x); c3 = self.layer2(c2); c4 = self.layer3(c3); c5 = self.layer4(c4)
        return [c2, c3, c4, c5]


@pytest.mark.parametrize("arch,block,layers,n_params", [("resnet18", _Basic, (2, 2, 2, 2), 11176512), ("resnet50", _Bottleneck, (3, 4, 6, 3), 23508032)])
@pytest.mark.parametrize("bn_mode", ["eval", "train"])
def test_image_trunk_equals_published_resnet(arch, block, layers, n_params, bn_mode):
    """The oracle's functional trunk == the module build, same state dict (strict load = identical key / shape list; the
    parameter counts are torchvision's resnet18 / resnet50 minus the 1000-way fc)."""
    shapes = {k: v for k, v in model_ref.image_state_shapes(64, arch=arch).items() if k.startswith("image_backbone.")}
    sd = model_ref.make_state_dict(shapes)
    net = _Trunk(block, layers)
    net.load_state_dict({k[len("image_backbone."):]: v for k, v in sd.items()}, strict=True)
    assert sum(p.numel() for p in net.parameters()) == n_params
    net.train(bn_mode == "train")
    g = torch.Generator().manual_seed(11)
    img = torch.randint(0, 256, (2, 3, 75, 131), dtype=torch.uint8, generator=g)     # odd sizes: every stride-2 rounding rule is exercised
    with torch.no_grad():
        ref = net(img.to(torch.float32) / 255.0)
        mine = model_ref.image_stream(sd, img, bn_mode, return_feats=True)
    for a, b in zip(mine, ref):
        assert a.shape == b.shape
        err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
        assert err < 1e-6, "%s %s: stage output differs by %g of its maximum" % (arch, bn_mode, err)


# ------------------------------------------------------------------------------------------------ KNN
def _cfg2_cloud():
    import bench
    import importlib
    cfg = bench.kitti_config(2)
    det = importlib.import_module(bench.PKG + ".detfill")
    calib = importlib.import_module(bench.PKG + ".calib")
    lim6 = (cfg["lidar_x_min"], cfg["lidar_x_max"], cfg["lidar_y_min"], cfg["lidar_y_max"], cfg["lidar_z_min"], cfg["lidar_z_max"])
    pts = det.synthetic_points(100000, lim6, 1234)
    _, pc, _, n, _ = geometry_ref.voxelization_projection(pts, cfg, calib.kitti_like_crt(), proj_mode="correct")
    return cfg, pc[:n]


def _check_against_kdtree(xyz, idx, X, Y, K):
    """idx [npix, K] from the oracle; (X, Y) [npix] the pixels' metric centres (fp32 values).  cKDTree works in float64 on the same
    fp32 inputs; the oracle orders by fp32 d2 = fl(fl(dx*dx) + fl(dy*dy)) then index.  The two may order a pair differently only
    where its float64 distances agree to fp32 rounding: everything else must be identical."""
    from scipy.spatial import cKDTree
    tree = cKDTree(xyz[:, :2].astype(np.float64))
    q = np.stack([X.astype(np.float64), Y.astype(np.float64)], 1)
    d_ref, i_ref = tree.query(q, k=K)
    same = (i_ref == idx)
    frac = same.all(1).mean()
    assert frac > 0.999, "only %.5f of the pixels have identical neighbour lists" % frac
    rows = np.nonzero(~same.all(1))[0]
    for r in rows:
        d_mine = np.sqrt(((xyz[idx[r], :2].astype(np.float64) - q[r]) ** 2).sum(1))
        # same distances rank by rank to fp32 rounding => a tie at that precision, resolved by index in the oracle
        assert np.allclose(d_mine, d_ref[r], rtol=3e-7, atol=0.0), (r, idx[r], i_ref[r], d_mine, d_ref[r])
    return frac, len(rows)


def test_knn_oracle_equals_ckdtree_at_cfg2_size():
    cfg, xyz = _cfg2_cloud()
    assert xyz.shape[0] > 30000
    g = geometry_ref.grid_constants(cfg)
    aff = g["aff"]
    K = 3
    # the whole stride-8 site (88 x 100) ...
    s, h, w = 8, cfg["voxel_length"] // 8, cfg["voxel_width"] // 8
    full = geometry_ref.knn_bev(xyz, K, h, w, s, aff, None)                       # [K, h, w]
    X = ((np.arange(h, dtype=np.float32) + np.float32(0.5)) * np.float32(s) - aff[1]) / aff[0]
    Y = ((np.arange(w, dtype=np.float32) + np.float32(0.5)) * np.float32(s) - aff[3]) / aff[2]
    XX, YY = np.meshgrid(X, Y, indexing="ij")
    _check_against_kdtree(xyz, full.reshape(K, -1).T, XX.ravel(), YY.ravel(), K)
    # ... and 6000 random pixels of the finest site (352 x 400, stride 2), K = 5 as in cfg4
    rng = np.random.RandomState(3)
    s, h, w, K = 2, cfg["voxel_length"] // 2, cfg["voxel_width"] // 2, 5
    pi = rng.randint(0, h, 6000).astype(np.int32)
    pj = rng.randint(0, w, 6000).astype(np.int32)
    some = geometry_ref.knn_pixels(xyz, K, pi, pj, s, aff, None)                  # [npix, K]
    X = ((pi.astype(np.float32) + np.float32(0.5)) * np.float32(s) - aff[1]) / aff[0]
    Y = ((pj.astype(np.float32) + np.float32(0.5)) * np.float32(s) - aff[3]) / aff[2]
    _check_against_kdtree(xyz, some, X, Y, K)
