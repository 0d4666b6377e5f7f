"""CPU suite: the oracle (oracle/) against the golden vectors generated from the imported
reference (oracle/gen_golden.py).  This is what pins the checker itself."""
import numpy as np
import pytest
import torch

from _util import golden_cfg, load_golden, pkg
from oracle import geometry_ref, loss_ref, model_ref


@pytest.mark.parametrize("case", ["two", "five", "n1k", "n10k"])
def test_geometry_oracle_bit_exact(case):
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    pts = z[case + "_pts"]
    grid, pc, uv, n, ids = geometry_ref.voxelization_projection(pts, cfg, z["crt"])
    assert n == int(z[case + "_n"])
    want = np.zeros(grid.size, np.float32)
    want[z[case + "_vox_idx"]] = z[case + "_vox_val"]
    assert np.array_equal(grid.reshape(-1).view(np.uint32), want.view(np.uint32))
    assert np.array_equal(uv[:n].view(np.uint32), z[case + "_uv"].view(np.uint32))
    assert np.array_equal(pc[:n].view(np.uint32), z[case + "_xyz"].view(np.uint32))
    assert np.array_equal(ids.astype(np.int16), z[case + "_ids"])
    assert not pc[n:].any() and not uv[n:].any()


@pytest.mark.parametrize("case", ["two", "five", "n1k", "n10k"])
def test_occupancy_grid_oracle_equals_reference(case):
    """interpolate=False (data_import_carla.py:231-234): the set of voxels the imported reference set to 1."""
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    g = geometry_ref.grid_constants(cfg)
    pin, _ = geometry_ref.range_filter(z[case + "_pts"], g["lim"])
    occ = geometry_ref.voxelize(pin, g["aff"], g["dims"], "occupancy")
    assert set(np.unique(occ)) <= {0.0, 1.0}
    assert np.array_equal(np.flatnonzero(occ.reshape(-1)).astype(np.int32), z[case + "_occ_idx"])


def test_voxel_last_writer_wins_not_accumulate():
    """five points in one voxel -> compat grid sums to 1.0 per corner family."""
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    g = geometry_ref.grid_constants(cfg)
    pin, _ = geometry_ref.range_filter(z["five_pts"], g["lim"])
    assert pin.shape[0] == 5
    compat = geometry_ref.voxelize(pin, g["aff"], g["dims"], "compat")
    accum = geometry_ref.voxelize(pin, g["aff"], g["dims"], "accum")
    assert abs(compat.sum() - 1.0) < 1e-5
    assert abs(accum.sum() - 5.0) < 1e-4


def test_knn_oracle_properties():
    z = load_golden("geometry_carla.npz")
    cfg = golden_cfg(z)
    g = geometry_ref.grid_constants(cfg)
    xyz = z["n1k_xyz"]
    K, h, w, s = 3, 24, 16, 16
    idx = geometry_ref.knn_bev(xyz, K, h, w, s, g["aff"])
    assert idx.shape == (K, h, w) and idx.min() >= 0 and idx.max() < xyz.shape[0]
    # independent numpy check with a stable sort on (d2, index)
    X = ((np.arange(h, dtype=np.float32) + np.float32(0.5)) * np.float32(s) - g["aff"][1]) / g["aff"][0]
    Y = ((np.arange(w, dtype=np.float32) + np.float32(0.5)) * np.float32(s) - g["aff"][3]) / g["aff"][2]
    dx = xyz[None, None, :, 0] - X[:, None, None]
    dy = xyz[None, None, :, 1] - Y[None, :, None]
    d2 = (dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)
    order = np.argsort(d2, axis=-1, kind="stable")[..., :K]
    assert np.array_equal(np.moveaxis(order, -1, 0).astype(np.int32), idx)
    # fewer candidates than K -> -1 padding ; rmax cuts
    few = geometry_ref.knn_bev(xyz[:2], K, h, w, s, g["aff"])
    assert (few[2] == -1).all() and (few[:2] >= 0).all()
    cut = geometry_ref.knn_bev(xyz, K, h, w, s, g["aff"], rmax=0.5)
    assert (cut == -1).any()
    assert geometry_ref.knn_bev(xyz[:0], K, h, w, s, g["aff"]).max() == -1


def test_anchors_and_decode():
    z = load_golden("anchors_decode.npz")
    cfg = golden_cfg(load_golden("geometry_carla.npz"))
    a = model_ref.anchors(cfg).numpy()
    assert np.array_equal(a.view(np.uint32), z["anchors_carla"].view(np.uint32))
    box = model_ref.decode(torch.from_numpy(z["reg"]), torch.from_numpy(z["anchors_tiny"])).numpy()
    assert np.array_equal(box, z["box"])


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_model_tiny_forward_backward(mode):
    z = load_golden("model_tiny.npz")
    cfg = golden_cfg(z)
    det = pkg("detfill")
    sd = model_ref.make_state_dict(model_ref.lidar_state_shapes(cfg))
    u = det.uniform((2, 32, 64, 32), 4242, 0.0, 1.0)
    m = det.uniform((2, 32, 64, 32), 4242 + 17, 0.0, 1.0) < 0.12
    x = torch.from_numpy((u * m).astype(np.float32))
    pred, st = model_ref.forward(sd, cfg, x, bn_mode=mode, return_stages=True)
    tol = 1e-4 * float(np.abs(z["pred_" + mode]).max())
    assert np.abs(pred.numpy() - z["pred_" + mode]).max() <= tol
    for k in ("layer2", "layer5", "fpn"):
        ref = z["stage_%s_%s" % (k, mode)]
        assert np.abs(st[k].numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    # backward through the restatement: d<pred,R>/d(input, weights)
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    xi = x[:1].clone().requires_grad_(True)
    out = model_ref.forward(params, cfg, xi, bn_mode=mode)
    R = torch.from_numpy(det.uniform((1, 32, 16, 8), 777, -1.0, 1.0))
    (out * R).sum().backward()
    assert np.abs(xi.grad.numpy() - z["gin_" + mode]).max() <= 2e-4 * np.abs(z["gin_" + mode]).max() + 1e-7
    for k in [str(s) for s in z["grad_keys"]]:
        ref = z["g_%s_%s" % (mode, k)]
        assert np.abs(params[k].grad.numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-6, k


def test_loss_restatement_seeded():
    z = load_golden("loss.npz")
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    anc = model_ref.anchors(cfg)
    for seed in (0, 1):
        cls = torch.from_numpy(z["cls"]).requires_grad_(True)
        reg = torch.from_numpy(z["reg"]).requires_grad_(True)
        np.random.seed(seed)
        val = loss_ref.loss_total(cfg, torch.from_numpy(z["bboxes"]), torch.from_numpy(z["nbox"]), cls, reg, anc)
        val.backward()
        assert abs(val.item() - float(z["loss_seed%d" % seed])) < 1e-6
        assert np.abs(cls.grad.numpy() - z["gcls_seed%d" % seed]).max() < 1e-7
        assert np.abs(reg.grad.numpy() - z["greg_seed%d" % seed]).max() < 1e-7
        assert float(cls.grad[0].abs().max()) == 0.0  # F5: only the last sample contributes


def test_full_carla_cfg1():
    """BASELINE configs[0]: 10k-pt CARLA frame through the (restated) reference model on CPU."""
    g = load_golden("geometry_carla.npz")
    z = load_golden("model_carla_full.npz")
    cfg = golden_cfg(g)
    grid, _, _, _, _ = geometry_ref.voxelization_projection(g["n10k_pts"], cfg, g["crt"])
    sd = model_ref.make_state_dict(model_ref.lidar_state_shapes(cfg))
    with torch.no_grad():
        pred = model_ref.forward(sd, cfg, torch.from_numpy(grid).unsqueeze(0), bn_mode="eval").numpy()[0]
    got = pred[:, z["sample_h"], z["sample_w"]]
    assert np.abs(got - z["sample_pred"]).max() <= 1e-4 * np.abs(z["sample_pred"]).max()
    assert np.allclose(pred.astype(np.float64).sum((1, 2)), z["chan_sum"], rtol=1e-4, atol=1e-2)


def _tiny_x():
    det = pkg("detfill")
    u = det.uniform((2, 32, 64, 32), 4242, 0.0, 1.0)
    m = det.uniform((2, 32, 64, 32), 4242 + 17, 0.0, 1.0) < 0.12
    return torch.from_numpy((u * m).astype(np.float32))


def test_quantisation_aware_statement_pinned_by_reference_golden():
    """oracle/model_quant_ref.py without rounding (qdtype=None) IS the fp32 network with the BatchNorm folded the way the
    device folds it: forward and every checked gradient equal the imported reference's golden vectors.  With rounding on,
    it moves away from them by a few storage ulps only (sanity of the rounding points)."""
    from oracle import model_quant_ref
    z = load_golden("model_tiny.npz")
    cfg = golden_cfg(z)
    det = pkg("detfill")
    sd = model_ref.make_state_dict(model_ref.lidar_state_shapes(cfg))
    x = _tiny_x()
    with torch.no_grad():
        pred = model_quant_ref.forward(sd, cfg, x, None).numpy()
    assert np.abs(pred - z["pred_eval"]).max() <= 1e-4 * np.abs(z["pred_eval"]).max()
    R = torch.from_numpy(det.uniform((1, 32, 16, 8), 777, -1.0, 1.0))
    grads = {}
    for dt in (None, torch.bfloat16, torch.float16):
        params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
        out = model_quant_ref.forward(params, cfg, x[:1].clone(), dt)
        (out * R).sum().backward()
        grads[dt] = {k: params[k].grad.numpy() for k in [str(s) for s in z["grad_keys"]]}
    for k, g in grads[None].items():
        ref = z["g_eval_" + k]
        assert np.abs(g - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-6, k
    for dt, lo, hi in ((torch.bfloat16, 1e-4, 2.5e-1), (torch.float16, 1e-5, 2.5e-1)):
        worst = max(np.abs(grads[dt][k] - grads[None][k]).max() / (np.abs(grads[None][k]).max() + 1e-12) for k in grads[None])
        assert lo < worst < hi, (dt, worst)
