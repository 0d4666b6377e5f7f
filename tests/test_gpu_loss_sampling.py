"""loss_sampling: device (/root/reference/loss.py:74-127 moved onto the device): the sampled lists against a
Python restatement of the sampler (bit for bit -- integer work), their properties (counts, window membership, rejection of the
selected positives, no entry chosen twice), and the loss / gradients against the list-driven kernel (dcf_loss_fwd_bwd, itself
pinned to the reference's golden numbers) fed with the same lists."""
import numpy as np
import pytest
import torch

from _util import M64, golden_cfg, load_golden, pkg, rand32

pytestmark = pytest.mark.gpu
def sampler_statement(L, boxes, nb, H, W, seed, sample):
    """The device sampler restated: (selected positive cells, negative cells, number of window entries)."""
    c = L.config
    f32 = np.float32
    rs, span = c["anchor_bbox_feature"]["reduced_scale"], c["positive_range"]
    half = span // 2
    entries = []
    for k in range(nb):
        cx = int((f32(boxes[k, 0]) * f32(L._xs) + f32(L._xo)) / f32(rs))
        cy = int((f32(boxes[k, 1]) * f32(L._ys) + f32(L._yo)) / f32(rs))
        if not (0 <= cx <= H - 1 and 0 <= cy <= W - 1):
            continue
        for dx in range(span):
            for dy in range(span):
                px, py = cx - half + dx, cy - half + dy
                if 0 <= px <= H - 1 and 0 <= py <= W - 1:
                    entries.append(px * W + py)
    cap = c["pos_sample_threshold"]
    if len(entries) > cap:
        keys = [(rand32(seed, sample, 1, i, 0), i) for i in range(len(entries))]
        chosen = sorted(i for _, i in sorted(keys)[:cap])
        pos = [entries[i] for i in chosen]
    else:
        pos = list(entries)
    taken = set(pos)
    neg = []
    for i in range(c["neg_sample_threshold"] + 1):
        att = 0
        while True:
            cell = (rand32(seed, sample, 2, i, att) * (H * W)) >> 32
            if cell not in taken:
                break
            att += 1
        neg.append(cell)
    return pos, neg, len(entries)


def _setup(n_boxes, seed=5, far=False):
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    cfg = dict(cfg, voxel_length=256, voxel_width=192, lidar_x_min=0.0, lidar_x_max=25.6, lidar_y_min=-9.6, lidar_y_max=9.6,
               loss_sampling="device", loss_seed=77, loss_reduction="mean")
    H, W = 64, 48
    g = torch.Generator().manual_seed(seed)
    B = 3
    boxes = torch.zeros(B, cfg["max_num_bbox"], 9)
    nb = []
    for b in range(B):
        n = n_boxes if b != 1 else max(n_boxes // 2, 0)
        for k in range(n):
            x = 1.0 + torch.rand(1, generator=g).item() * 23.0
            y = -9.0 + torch.rand(1, generator=g).item() * 18.0
            if far and k == 0:
                x, y = 40.0, 3.0                        # outside the grid: no window
            if k == 1:
                x, y = 0.05, -9.55                      # corner: a clipped window
            boxes[b, k] = torch.tensor([x, y, -1.0, 4.0, 1.8, 1.5, 0.3 * k, 6, 1])
        nb.append(n)
    cls = torch.rand(B, 4, H, W, generator=g)
    reg = torch.rand(B, 14, H, W, generator=g) - 0.5
    return cfg, boxes, torch.tensor(nb), cls, reg, H, W


@pytest.mark.parametrize("n_boxes", [0, 3, 20])
def test_device_sampler_lists_properties_and_loss(n_boxes):
    cfg, boxes, nb, cls, reg, H, W = _setup(n_boxes, far=n_boxes >= 3)
    Lm = pkg("loss")
    L = Lm.LossTotal(cfg).cuda()
    L.keep_samples = True
    c1 = cls.cuda().requires_grad_(True)
    r1 = reg.cuda().requires_grad_(True)
    loss = L(boxes, nb, c1, r1)
    loss.backward()
    pos, neg, counts = [t.cpu().numpy() for t in L.last_samples]
    seed = (cfg["loss_seed"] * 0x9E3779B1 + 0) & M64
    cap, nneg = cfg["pos_sample_threshold"], cfg["neg_sample_threshold"] + 1
    span = cfg["positive_range"]
    ints, floats, plan = [], [], []
    Lc = Lm.LossTotal(dict(cfg, loss_sampling="compat"))
    for b in range(boxes.shape[0]):
        n = int(nb[b])
        want_pos, want_neg, n_entries = sampler_statement(L, boxes[b].numpy(), n, H, W, seed, b)
        got_pos = [int(v) for v in pos[b] if v >= 0]
        # --- bit-exact lists
        assert got_pos == want_pos and [int(v) for v in neg[b]] == want_neg
        assert int(counts[b, 0]) == len(want_pos) == min(n_entries, cap) and int(counts[b, 1]) == n_entries
        assert list(pos[b][len(got_pos):]) == [-1] * (cap - len(got_pos))
        # --- properties
        assert len(want_neg) == nneg and not (set(want_neg) & set(want_pos))
        assert all(0 <= v < H * W for v in want_neg)
        cells = set()
        for k in range(n):
            cx = int((np.float32(boxes[b, k, 0]) * np.float32(L._xs) + np.float32(L._xo)) / np.float32(4))
            cy = int((np.float32(boxes[b, k, 1]) * np.float32(L._ys) + np.float32(L._yo)) / np.float32(4))
            if 0 <= cx < H and 0 <= cy < W:
                cells |= set(px * W + py for px in range(max(cx - span // 2, 0), min(cx + span // 2, H - 1) + 1)
                             for py in range(max(cy - span // 2, 0), min(cy + span // 2, W - 1) + 1))
        assert set(want_pos) <= cells
        if n_entries > cap:
            assert len(want_pos) == cap
        # --- the same lists through the list-driven kernel
        np.random.seed(0)
        _, _, regress, owner = Lc.assign(boxes[b, :n], H, W)
        rows, row_box, row_w = [], [], []
        for k in range(n):
            for m in owner[k]:
                rows.append(regress[m][0] * W + regress[m][1]); row_box.append(k); row_w.append(1.0 / (len(owner[k]) * 14))
        o = len(ints)
        ints += want_pos + want_neg + rows + row_box
        of = len(floats)
        floats += row_w + boxes[b, :n, :7].reshape(-1).tolist()
        plan.append((o, len(want_pos), len(want_neg), len(rows), of, n))
    c2 = cls.cuda().requires_grad_(True)
    r2 = reg.cuda().requires_grad_(True)
    Lc = Lc.cuda()
    anc = Lc.anchor_set.cuda().reshape(2, 7, H * W)
    ref = Lc._forward_hip(c2, r2, anc, ints, floats, plan, boxes.shape[0], H, W)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert torch.allclose(c1.grad, c2.grad, rtol=1e-5, atol=1e-7) and torch.allclose(r1.grad, r2.grad, rtol=1e-5, atol=1e-7)
    if n_boxes == 20:
        assert int(counts[0, 1]) > cap                   # the subset branch really ran


def test_device_sampler_is_stateless_and_advances_per_call():
    cfg, boxes, nb, cls, reg, H, W = _setup(6)
    Lm = pkg("loss")
    out = []
    for rep in range(2):
        L = Lm.LossTotal(cfg).cuda()
        L.keep_samples = True
        lists = []
        for step in range(2):
            L(boxes, nb, cls.cuda(), reg.cuda())
            lists.append([t.cpu().clone() for t in L.last_samples])
        out.append(lists)
    for step in range(2):
        assert all(torch.equal(a, b) for a, b in zip(out[0][step], out[1][step]))      # same seed, same call index: same lists
    assert not torch.equal(out[0][0][1], out[0][1][1])                                  # the next call draws other negatives
    # labels already on the device, counts as a tensor: same lists
    L = Lm.LossTotal(cfg).cuda()
    L.keep_samples = True
    L(boxes.cuda(), nb.cuda(), cls.cuda(), reg.cuda())
    assert all(torch.equal(a.cpu(), b) for a, b in zip(L.last_samples, out[0][0]))


def test_device_sampling_needs_the_device():
    cfg, boxes, nb, cls, reg, H, W = _setup(2)
    L = pkg("loss").LossTotal(cfg)
    with pytest.raises(RuntimeError):
        L(boxes, nb, cls, reg)


def test_train_step_with_device_sampling():
    """Train.one_step with loss_sampling: device on the tiny model: finite loss, parameters move, no host RNG consumed."""
    z = load_golden("model_tiny.npz")
    lz = load_golden("loss.npz")
    cfg = golden_cfg(z)
    cfg.update(dtype="f32", loss_sampling="device", loss_seed=3)
    T = pkg("train")
    det = pkg("detfill")
    tr = T.Train(cfg)
    det.fill_state_dict(tr.model)
    u = det.uniform((1, 32, 64, 32), 4242, 0.0, 1.0)
    x = torch.from_numpy(u.astype(np.float32)).cuda()
    img = torch.zeros(1, 3, 8, 8, dtype=torch.uint8, device="cuda")
    boxes, nbx = torch.from_numpy(lz["bboxes"])[:1], torch.from_numpy(lz["nbox"])[:1]
    before = tr.model.flat_params.clone()
    np.random.seed(11)
    state = np.random.get_state()[1].copy()
    for _ in range(2):
        tr.one_step(x, img, boxes, nbx)
    assert np.array_equal(np.random.get_state()[1], state)
    assert np.isfinite(tr.loss_value.item()) and not torch.equal(before, tr.model.flat_params)
