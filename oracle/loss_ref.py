"""TEST INFRASTRUCTURE -- torch-CPU restatement of the reference objective.

Follows loss.py:33-189 (LossTotal).  PINNED: seeded scalars and gradients are
checked against the imported reference in tests/golden/loss.npz.

Quirks kept on purpose: CE applied to already-softmaxed
scores (loss.py:17-20,139), 129 negatives (`>` at :125), only the LAST sample
of the batch contributes (:71 overwrites), numpy global RNG drives the
sampling (:108,:118-119) so np.random.seed() pins it.
"""
import numpy as np
import torch
import torch.nn.functional as F


def sample_positions(cfg, boxes, H, W):
    """loss.py:74-127 -- host-side target assignment; consumes np.random exactly
    like the reference (one shuffle of the window list, then randint pairs)."""
    L, Wd = cfg["voxel_length"], cfg["voxel_width"]
    xs = int(L / (cfg["lidar_x_max"] - cfg["lidar_x_min"]))
    ys = int(Wd / (cfg["lidar_y_max"] - cfg["lidar_y_min"]))
    xo = int(-cfg["lidar_x_min"] * xs)
    yo = int(-cfg["lidar_y_min"] * ys)
    rs = cfg["anchor_bbox_feature"]["reduced_scale"]
    pr = cfg["positive_range"]
    half = int(pr / 2)
    pos, reg_pos, groups = [], [], {}
    cnt = 0
    for bi, box in enumerate(boxes):
        groups[bi] = []
        px = int((box[0] * xs + xo) / rs)
        py = int((box[1] * ys + yo) / rs)
        if px < 0 or px > H - 1 or py < 0 or py > W - 1:
            continue
        for ix in range(pr):
            qx = px - half + ix
            for iy in range(pr):
                qy = py - half + iy
                if qx < 0 or qx > H - 1 or qy < 0 or qy > W - 1:
                    continue
                pos.append([qx, qy])
                if cfg["regress_type"] == 0 or (qx == px and qy == py):
                    reg_pos.append([qx, qy])
                    groups[bi].append(cnt)
                    cnt += 1
    np.random.shuffle(pos)
    if len(pos) > cfg["pos_sample_threshold"]:
        pos = pos[:cfg["pos_sample_threshold"]]
    neg = []
    while True:
        x = np.random.randint(H)
        y = np.random.randint(W)
        if [x, y] in pos:
            continue
        neg.append([x, y])
        if len(neg) > cfg["neg_sample_threshold"]:
            break
    return pos, neg, reg_pos, groups


def _class_term(pos, neg, score2):
    """loss.py:129-142: CrossEntropy (mean) on the 2-way scores gathered at pos / neg."""
    n_idx = torch.tensor(neg, dtype=torch.long)
    c = score2[:, n_idx[:, 0], n_idx[:, 1]].permute(1, 0)
    out = F.cross_entropy(c, torch.zeros(len(neg), dtype=torch.long))
    if len(pos) > 0:
        p_idx = torch.tensor(pos, dtype=torch.long)
        a = score2[:, p_idx[:, 0], p_idx[:, 1]].permute(1, 0)
        out = F.cross_entropy(a, torch.ones(len(pos), dtype=torch.long)) + out
    return out


def _reg_term(box, pred, anc):
    """loss.py:144-165.  box [>=7]; pred [N,14]; anc [N,2,7] -> mean Smooth-L1 of encoded offsets."""
    N = anc.shape[0]
    ref = box[:7].view(1, 1, 7).expand(N, 2, 7)
    p = pred.reshape(N, 2, 7)
    diag = torch.sqrt(torch.pow(anc[:, :, 3:4], 2) + torch.pow(anc[:, :, 4:5], 2))
    t_xy = (ref[:, :, 0:2] - anc[:, :, 0:2]) / diag
    t_z = (ref[:, :, 2:3] - anc[:, :, 2:3]) / anc[:, :, 5:6]
    t_lwh = torch.log(ref[:, :, 3:6] / anc[:, :, 3:6])
    d = ref[:, :, 6] - anc[:, :, 6]
    t_yaw = torch.atan2(torch.sin(d), torch.cos(d)).unsqueeze(-1)
    tgt = torch.cat((t_xy, t_z, t_lwh, t_yaw), -1)
    return F.smooth_l1_loss(p, tgt, reduction="none").sum() * (1.0 / (N * 2 * 7))


def loss_total(cfg, bboxes, nbox, cls, reg, anc14, reduction="last"):
    """loss.py:46-72.  bboxes [B,max,9]; nbox [B]; cls [B,4,h,w]; reg [B,14,h,w]; anc14 [14,h,w].

    reduction 'last' = reference behaviour (F5); 'sum' / 'mean' accumulate over the batch.
    Returns a [1] tensor.
    """
    B = bboxes.shape[0]
    H, W = cls.shape[-2:]
    anc = anc14.reshape(2, 7, H, W)
    total = torch.zeros(1)
    acc = torch.zeros(1)
    for b in range(B):
        boxes = bboxes[b, :int(nbox[b])]
        pos, neg, reg_pos, groups = sample_positions(cfg, boxes, H, W)
        lc = _class_term(pos, neg, cls[b, 0:2]) + _class_term(pos, neg, cls[b, 2:4])
        lr = torch.zeros(1)
        reg_pos_arr = np.array(reg_pos)
        for gi, box in enumerate(boxes):
            sel = torch.tensor(reg_pos_arr[groups[gi]], dtype=torch.long) if len(groups[gi]) else None
            if sel is None or sel.numel() == 0:
                continue
            pb = reg[b][:, sel[:, 0], sel[:, 1]].permute(1, 0)
            ab = anc[:, :, sel[:, 0], sel[:, 1]].permute(2, 0, 1)
            lr = lr + _reg_term(box, pb, ab)
        total = lc + cfg["regress_loss_gain"] * lr
        acc = acc + total
    if reduction == "last":
        return total
    if reduction == "sum":
        return acc
    return acc / B
