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


def test_product_loss_cell_boundary_matches_reference():
    """A box centre within fp32 rounding of a cell boundary (tests/golden/loss_boundary.npz, made by the imported
    reference): the positive window must sit where the reference's fp32 tensor arithmetic puts it (loss.py:85-86), one
    cell above what the same expression gives in double precision."""
    z = load_golden("loss_boundary.npz")
    assert not np.array_equal(z["cell_f32"], z["cell_f64"])
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    L = pkg("loss").LossTotal(cfg)
    cls = torch.from_numpy(z["cls"]).requires_grad_(True)
    reg = torch.from_numpy(z["reg"]).requires_grad_(True)
    np.random.seed(5)
    val = L(torch.from_numpy(z["bboxes"]), torch.from_numpy(z["nbox"]), cls, reg)
    val.backward()
    assert abs(val.item() - float(z["loss"])) < 1e-6
    assert np.abs(cls.grad.numpy() - z["gcls"]).max() < 1e-7
    assert np.abs(reg.grad.numpy() - z["greg"]).max() < 1e-7


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


def test_eval_postprocessing_matches_reference_golden():
    """NMS_IOU / NMS_SAT survivors, precision-recall counters and the pairwise IoUs of tests/golden/eval.npz (produced
    by the imported reference's Test / IOU.py) -- and the known answers printed in IOU.py:161-167."""
    z = load_golden("eval.npz")
    EG = pkg("evalgeom")
    g = EG.box_corners((2.882992, 1.698800, 20.785644), (1.497255, 1.644981, 3.628938), -1.531692)
    p = EG.box_corners((2.756923, 1.661275, 20.943280), (1.458242, 1.604773, 3.707947), -1.549553)
    assert np.abs(np.array(EG.rotated_iou(p, g)) - z["known_iou"]).max() < 1e-12
    assert abs(z["known_iou"][0] - 0.68796) < 1e-5 and abs(z["known_iou"][1] - 0.70037) < 1e-5
    c = [EG.box_corners(x[:3], x[3:6], x[6]) for x in z["pred"][0][:12].astype(np.float64)]
    got3 = np.array([[EG.rotated_iou(c[i], c[j])[0] for j in range(12)] for i in range(12)])
    got2 = np.array([[EG.rotated_iou(c[i], c[j])[1] for j in range(12)] for i in range(12)])
    off = ~np.eye(12, dtype=bool)
    assert np.abs(got3 - z["pair_iou3d"])[off].max() < 1e-9 and np.abs(got2 - z["pair_iou2d"])[off].max() < 1e-9
    # a box against ITSELF is degenerate in the reference (every vertex sits on a clip edge of the strict half-plane test:
    # its fixture values range from -11.2 to 3.5); this implementation returns exactly 1
    assert np.abs(np.diag(got3) - 1).max() < 1e-12 and not (np.abs(np.diag(z["pair_iou3d"]) - 1) < 1e-6).all()
    a, b, c3 = [(0, 0), (70, 70), (70, 0), (0, 70)], [(70, 70), (150, 70), (150, 150), (70, 150)], [(30, 30), (150, 70), (70, 150)]
    assert EG.rects_overlap(a, b) and EG.rects_overlap(a, c3) and EG.rects_overlap(b, c3)      # separation_axis_theorem.py main()
    assert not EG.rects_overlap([(0, 0), (1, 0), (1, 1), (0, 1)], [(2, 0), (3, 0), (3, 1), (2, 1)])

    T = pkg("test").Test.__new__(pkg("test").Test)
    T.initialize_ap()
    pred = [torch.from_numpy(z["pred"][b]) for b in range(2)]

    def idx(kept, boxes):
        return np.array([int(np.where((boxes.numpy() == k.numpy()).all(1))[0][0]) for k in kept], dtype=np.int64)
    ki, ks = T.NMS_IOU(pred, 0.01), T.NMS_SAT(pred)
    for b in range(2):
        assert np.array_equal(idx(ki[b], pred[b]), z["keep_iou_%d" % b])
        assert np.array_equal(idx(ks[b], pred[b]), z["keep_sat_%d" % b])
    T.precision_recall_singleshot(ks, torch.from_numpy(z["ref"]))
    assert T.get_num_T() == int(z["num_T"]) and T.get_num_P() == int(z["num_P"])
    assert [T.get_num_TP_set()[t] for t in T.IOU_threshold] == z["num_TP"].tolist()
    prec, rec = T.display_average_precision()
    assert len(prec[0.5]) == T.get_num_P() + 1 and rec[0.5][-1] == T.get_num_TP_set()[0.5] / T.get_num_T()
    # edge cases: no predictions, no labels
    assert T.NMS_SAT([torch.zeros(0, 7)]) == [[]] and T.NMS_IOU([torch.zeros(0, 7)]) == [[]]
    T.initialize_ap()
    T.precision_recall_singleshot([[]], torch.zeros(1, 20, 9))
    assert T.get_num_T() == 0 and T.get_num_P() == 0


def test_batched_negative_sampling_consumes_the_generator_like_scalar_calls():
    """loss.LossTotal.assign draws its negative cells in batches; the legacy numpy generator must be left exactly where
    the reference's scalar loop (loss.py:117-126: randint(H), randint(W), reject if positive) leaves it."""
    import numpy as np
    for H, W, seed in ((176, 200, 0), (96, 64, 5), (16, 8, 11), (4, 4, 3)):
        taken = {(i % H, (3 * i) % W) for i in range(0, min(H * W // 2, 128))}
        np.random.seed(seed)
        ref = []
        while len(ref) <= 128 and len(ref) < H * W - len(taken):
            cand = (np.random.randint(H), np.random.randint(W))
            if cand in taken:
                continue
            ref.append(cand)
        tail_ref = np.random.randint(1 << 30)
        np.random.seed(seed)
        got, want = [], len(ref)
        while len(got) < want:
            for x, y in np.random.randint([H, W], size=(want - len(got), 2)).tolist():
                if (x, y) not in taken:
                    got.append((x, y))
        assert got == ref and np.random.randint(1 << 30) == tail_ref



def test_assign_arrays_equals_assign_lists_and_generator_state():
    """LossTotal.assign_arrays (numpy, the CUDA path of the compat mode) against LossTotal.assign (lists; pinned to the reference's
    golden loss above): same positive / negative / regression cells in the same order, same weights, and numpy's legacy
    generator left in the same state -- over box counts from 0 to the maximum, clipped windows, boxes outside the grid, more
    window cells than pos_sample_threshold (the subset branch), both regress types."""
    import numpy as np
    from _util import golden_cfg, load_golden, pkg
    cfg0 = golden_cfg(load_golden("model_tiny.npz"))
    rng = np.random.default_rng(3)
    for regress_type in (0, 1):
        cfg = dict(cfg0, voxel_length=704, voxel_width=800, lidar_x_min=0.0, lidar_x_max=70.4, lidar_y_min=-40.0, lidar_y_max=40.0,
                   regress_type=regress_type)
        L = pkg("loss").LossTotal(cfg)
        H, W = 176, 200
        for trial, nb in enumerate((0, 1, 3, 8, 20, 20)):
            boxes = np.zeros((nb, 9), np.float32)
            boxes[:, 0] = rng.uniform(-3.0, 75.0, nb)          # some centres outside the grid
            boxes[:, 1] = rng.uniform(-42.0, 42.0, nb)
            if nb >= 3:
                boxes[1, :2] = (0.05, -39.95)                  # a corner: clipped window
                boxes[2, :2] = (35.2, 0.0)                     # exactly on a cell boundary
            if trial == 5:
                boxes[:, 0] = rng.uniform(5.0, 65.0, nb); boxes[:, 1] = rng.uniform(-35.0, 35.0, nb)    # all inside: 500 cells > 128
            tb = torch.from_numpy(boxes)
            np.random.seed(40 + trial)
            pos, neg, regress, owner = L.assign(tb, H, W)
            tail = np.random.randint(1 << 30)
            np.random.seed(40 + trial)
            apos, aneg, rows, row_box, row_w = L.assign_arrays(boxes, H, W)
            assert np.random.randint(1 << 30) == tail
            assert apos.tolist() == [p[0] * W + p[1] for p in pos]
            assert aneg.tolist() == [q[0] * W + q[1] for q in neg]
            want_rows, want_box, want_w = [], [], []
            for k in range(nb):
                for m in owner[k]:
                    want_rows.append(regress[m][0] * W + regress[m][1]); want_box.append(k); want_w.append(1.0 / (len(owner[k]) * 14))
            assert rows.tolist() == want_rows and row_box.tolist() == want_box
            assert np.allclose(row_w, np.array(want_w, np.float32), rtol=0, atol=0)
            if trial == 5:
                assert len(pos) == cfg["pos_sample_threshold"]


def test_profile_names_map_to_kernels_of_the_committed_pmc_summary():
    """bench.rocprof_kernel turns the runtime's launch names into rocprofv3's (function, template arguments); a name it maps
    wrongly silently loses its `traffic` figure, so the families of the cfg2 step are pinned to the committed summary."""
    import csv
    import glob
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    # the newest cfg2 summary (rNNx_pmc_traffic.csv; the cfg4 / cfg5 ones carry a tag in their names)
    files = bench.pmc_summaries("")              # (oldest first; r05z < r05zb < r05zc)
    assert files and [os.path.basename(f)[:5] for f in files if os.path.basename(f).startswith("r05z")][:3] == ["r05z_", "r05zb", "r05zc"][:len([f for f in files if os.path.basename(f).startswith("r05z")])]
    legacy = int(os.path.basename(files[-1])[1:3]) < 5          # before round 5 k_conv3x3_rs had no CHAIN template argument
    have = set()
    for row in csv.DictReader(open(files[-1])):
        k = row["kernel"]
        if ">(" not in k:
            continue
        head = k.split(">(")[0]
        have.add((head[:head.index("<")].split("::")[-1].split()[-1], tuple(head[head.index("<") + 1:].replace(" ", "").split(","))))
    # (round 5: a summary collected with the opt-in chain launches carries the CHAIN = true instantiations, and the 128-channel stage's
    # single-layer instantiation is then not in a cfg2 step; the default tree runs the single-layer ones)
    chained = any(f == "k_conv3x3_rs" and a[-1] == "true" and len(a) == 9 for f, a in have)
    # (since the end of round 5 every kind-1 launch of a cfg2 step has at most eight position tiles and runs the DX = 2 instantiation)
    dx2 = any(f == "k_conv3x3_rs" and a[1:7] == ("1", "2", "2", "4", "2", "2") for f, a in have)
    names = ["conv_fwd_bf16<sp32>", "conv_dgrad_bf16<sp32>", "conv_fwd_bf16<sp64>", "conv_fwd_bf16<rs1,7>" if dx2 else "conv_fwd_bf16<rs1,9>", "conv_fwd_bf16<rs2,5>"]
    names += ["conv_fwd_bf16<rs0,9,x7>", "conv_dgrad_bf16<rs1,7,x11>", "conv_fwd_bf16<rs2,3,x11>"] if chained else ["conv_fwd_bf16<rs0,9>"]
    # (summaries collected before the 16-wave form of the small-M kind existed -- up to r05w -- carry nine template arguments)
    # (... and summaries of round 5 ten: the eleventh, the rotated tap loop, is round 6's)
    nargs = max((len(a) for f, a in have if f == "k_conv3x3_rs"), default=11)
    pf_seen = any(f == "k_conv3x3_rs" and len(a) == 11 and a[10] == "true" for f, a in have)
    for name in names:
        if pf_seen and name in ("conv_fwd_bf16<rs0,9>", "conv_fwd_bf16<rs1,7>", "conv_fwd_bf16<rs2,5>"):
            name = name[:-1] + ",pf>"                  # the 128+ channel layers and the small-M kind of a cfg2 step run the rotated loop
        func, args = bench.rocprof_kernel(name)
        if func == "k_conv3x3_rs":
            args = args[:nargs]
        if legacy and func == "k_conv3x3_rs":
            assert args[-1] == "false"
            args = args[:-1]
        assert (func, tuple(args)) in have, (name, func, args)
    assert bench.rocprof_kernel("conv_dgrad_bf16<rs2,4,x21>") == ("k_conv3x3_rs", ["unsignedshort", "1", "1", "2", "4", "6", "2", "true", "true", "0", "false"])
    assert bench.rocprof_kernel("conv_fwd_bf16<rs2,3>") == ("k_conv3x3_rs", ["unsignedshort", "1", "1", "2", "4", "6", "2", "true", "false", "8", "false"])
    assert bench.rocprof_kernel("conv_fwd_bf16<rs2,3,pf>") == ("k_conv3x3_rs", ["unsignedshort", "1", "1", "2", "4", "6", "2", "true", "false", "8", "true"])
    assert bench.rocprof_kernel("conv_fwd_bf16<rs1,7,pf>") == ("k_conv3x3_rs", ["unsignedshort", "1", "2", "2", "4", "2", "2", "false", "false", "0", "true"])
    assert bench.rocprof_kernel("conv_dgrad_bf16<rs0,9,pf>") == ("k_conv3x3_rs", ["unsignedshort", "1", "5", "4", "2", "2", "1", "false", "false", "0", "true"])


def test_lidar_backbone_network_needs_no_config():
    """/root/reference/model.py:139: `LidarBackboneNetwork()` takes no config; the parameters do not depend on the BEV grid."""
    import importlib
    m = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.model")
    net = m.LidarBackboneNetwork()
    assert sum(p.numel() for p in net.parameters()) == 12599040            # the reference's count
    small = m.LidarBackboneNetwork(out_feature=(32, 64, 96, 128, 160), num_res_block=(1, 1, 2, 1, 1))
    keys = set(small.net.state_dict().keys())
    assert any(k.endswith("classconv.weight") for k in keys) and any(k.endswith("bbox3dconv.weight") for k in keys)


def test_lidar_backbone_network_replans_in_place():
    """a re-plan for another grid must keep the module, its arenas and every nn.Parameter OBJECT -- an optimizer
    created before the first forward (the reference's order: model, optimizer, forward; train.py:23-28) holds those objects."""
    import importlib
    import pytest
    m = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.model")
    net = m.LidarBackboneNetwork(out_feature=(32, 64, 96, 128, 160), num_res_block=(1, 1, 2, 1, 1))
    inner = net.net
    before = [(id(p), p.data_ptr(), p.requires_grad) for p in net.parameters()]
    list(net.parameters())[3].requires_grad_(False)
    before[3] = (before[3][0], before[3][1], False)
    inner.set_grid(64, 32)
    assert net.net is inner and (inner.config["voxel_length"], inner.config["voxel_width"]) == (64, 32)
    assert [(id(p), p.data_ptr(), p.requires_grad) for p in net.parameters()] == before
    anc = m.AnchorBoundingBoxFeature(inner.config)()
    assert tuple(anc.shape) == (14, 16, 8)
    with pytest.raises(ValueError):
        inner.set_grid(70, 32)                      # not a multiple of 16 (model.py:151)
    assert (inner.config["voxel_length"], inner.config["voxel_width"]) == (64, 32)


def test_num_batches_tracked_counters_are_views_of_one_tensor():
    """Round 5: every BatchNorm's num_batches_tracked (reference state_dict key, 0-d int64) is a view of ONE flat tensor, so a
    train-mode forward bumps them with one launch; state_dict round trips and .to() keep the views bound."""
    import importlib
    import torch
    m = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.model")
    net = m.LidarBackboneNetwork()
    inner = getattr(net, "net", net)
    sd = inner.state_dict()
    keys = [k for k in sd if k.endswith("num_batches_tracked")]
    assert len(keys) == len(inner._nbt_keys) > 10
    assert all(sd[k].shape == () and sd[k].dtype == torch.long for k in keys)
    inner._nbtflat += 3
    assert all(int(v) == 3 for k, v in inner.state_dict().items() if k.endswith("num_batches_tracked"))
    sd2 = {k: (torch.tensor(7) if k.endswith("num_batches_tracked") else v.clone()) for k, v in inner.state_dict().items()}
    inner.load_state_dict(sd2)
    assert int(inner._nbtflat.min()) == 7 and int(inner._nbtflat.max()) == 7
    inner.to("cpu")                               # same device: the arenas stay, the views stay bound
    inner._nbtflat += 1
    assert all(int(v) == 8 for k, v in inner.state_dict().items() if k.endswith("num_batches_tracked"))
