"""CPU suite: host-side logic of the product (no kernels): loss target assignment against the
golden vectors, plan/parameter-table construction, config handling, calibration constants."""
import copy

import numpy as np
import pytest
import torch

from _util import golden_cfg, load_golden, pkg


def test_product_loss_matches_reference_golden():
    z = load_golden("loss.npz")
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    L = pkg("loss").LossTotal(cfg)
    for seed in (0, 1):
        cls = torch.from_numpy(z["cls"]).requires_grad_(True)
        reg = torch.from_numpy(z["reg"]).requires_grad_(True)
        np.random.seed(seed)
        val = L(torch.from_numpy(z["bboxes"]), torch.from_numpy(z["nbox"]), cls, reg)
        val.backward()
        assert abs(val.item() - float(z["loss_seed%d" % seed])) < 1e-6
        assert np.abs(cls.grad.numpy() - z["gcls_seed%d" % seed]).max() < 1e-7
        assert np.abs(reg.grad.numpy() - z["greg_seed%d" % seed]).max() < 1e-7
    cfg2 = copy.deepcopy(cfg)
    cfg2["loss_reduction"] = "sum"
    np.random.seed(0)
    s = pkg("loss").LossTotal(cfg2)(torch.from_numpy(z["bboxes"]), torch.from_numpy(z["nbox"]), torch.from_numpy(z["cls"]),
                                    torch.from_numpy(z["reg"]))
    assert s.item() > float(z["loss_seed0"])          # both samples now contribute


def test_anchor_surface_bit_exact():
    z = load_golden("anchors_decode.npz")
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    anc = pkg("model").AnchorBoundingBoxFeature(cfg)().numpy()
    assert np.array_equal(anc.view(np.uint32), z["anchors_carla"].view(np.uint32))


def test_parameter_table_and_views():
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    net = pkg("model").ObjectDetection_DCF(cfg)
    assert sum(p.numel() for p in net.parameters()) == 12599040
    assert len(net.state_dict()) == 258
    # parameters are views of one flat arena, convolution weights physically [O,kh,kw,I]
    w = dict(net.named_parameters())["lidar_backbone.conv3.weight"]
    assert w.shape == (192, 192, 3, 3) and w.stride() == (192 * 9, 1, 3 * 192, 192)
    assert w.data_ptr() >= net.flat_params.data_ptr()
    w.data.fill_(2.0)
    assert float(net.flat_params.max()) == 2.0
    # fused heads: classconv and bbox3dconv are adjacent rows of one [18][C] matrix
    a = dict(net.named_parameters())["lidar_backbone.classconv.weight"]
    b = dict(net.named_parameters())["lidar_backbone.bbox3dconv.weight"]
    assert b.data_ptr() - a.data_ptr() == 4 * 192 * 4


def test_plan_rejects_bad_configs():
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    M = pkg("model")
    bad = copy.deepcopy(cfg); bad["voxel_length"] = 700
    with pytest.raises(ValueError):
        M.ObjectDetection_DCF(bad)
    bad = copy.deepcopy(cfg); bad["voxel_channel"] = 16
    with pytest.raises(ValueError):
        M.ObjectDetection_DCF(bad)
    bad = copy.deepcopy(cfg); bad["bn_mode"] = "sync"
    with pytest.raises(ValueError):
        M.ObjectDetection_DCF(bad)


def test_fused_model_state_dict_names():
    from oracle import model_ref
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    for arch in ("resnet18", "resnet50"):
        cfg["fusion"] = {"enabled": True, "K": 3, "image_channels": 64, "image_stream": arch}
        net = pkg("model").ObjectDetection_DCF(cfg)
        want = {}
        want.update(model_ref.lidar_state_shapes(cfg))
        want.update(model_ref.image_state_shapes(64, arch=arch))
        want.update(model_ref.fusion_state_shapes(cfg, 64))
        got = net.state_dict()
        assert set(got.keys()) == set(want.keys()), arch
        assert all(tuple(got[k].shape) == tuple(want[k]) for k in want), arch
    # torchvision's resnet50 has 23,508,032 trunk parameters without the fc layer (53 convs + 53 BN pairs)
    n50 = sum(int(np.prod(v)) for k, v in model_ref.image_state_shapes(64, arch="resnet50").items()
              if k.startswith("image_backbone") and "running" not in k and "num_batches" not in k)
    assert n50 == 23508032, n50


def test_calibration_matches_golden_crt():
    z = load_golden("geometry_carla.npz")
    assert np.abs(pkg("calib").carla_crt() - z["crt"]).max() < 1e-4
    k = pkg("calib").kitti_like_crt()
    assert k.shape == (4, 3) and k.dtype == np.float32 and not k[3].any()


def test_gridspec_truncation_semantics():
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    g = pkg("ops").GridSpec(cfg)
    assert (g.xs, g.ys, g.zs, g.xo, g.yo, g.zo) == (5, 4, 10, 0, 120, 24)
    assert g.lim.dtype == np.float32 and g.lim[1] == np.float32(69.8)
