"""Drop-in counterpart of the reference's loss.py (LossTotal) -- the train-step harness row.

Same constructor and call surface: LossTotal(config)(bboxes [B,max,9], num_boxes [B],
cls [B,4,h,w], reg [B,14,h,w]) -> [1] loss tensor.  Target assignment is host-side Python
driven by numpy's global RNG exactly like loss.py:74-127 (so np.random.seed pins it); the
gathers / cross-entropy / Smooth-L1 are a handful of tiny torch ops on the model's device
(harness, not a HIP target this round; "next" row N1).

Reference quirks are kept behind `loss_reduction: last` (default): cross-entropy on already
soft-maxed scores (loss.py:17-20,139), 129 negatives (:125), only the last sample of the
batch contributes (:71).  'sum' / 'mean' accumulate over the batch instead.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import AnchorBoundingBoxFeature


class LossTotal(nn.Module):
    def __init__(self, config):
        super(LossTotal, self).__init__()
        self.config = config
        self.regress_type = config["regress_type"]
        self.reduction = config.get("loss_reduction", "last")
        anc = AnchorBoundingBoxFeature(config)()
        self.register_buffer("anchor_set", anc.reshape(2, 7, anc.shape[1], anc.shape[2]), persistent=False)
        L, W = config["voxel_length"], config["voxel_width"]
        self._xs = int(L / (config["lidar_x_max"] - config["lidar_x_min"]))
        self._ys = int(W / (config["lidar_y_max"] - config["lidar_y_min"]))
        self._xo = int(-config["lidar_x_min"] * self._xs)
        self._yo = int(-config["lidar_y_min"] * self._ys)

    # ------------------------------------------------------------------ host-side sampling
    def assign(self, boxes, H, W):
        """Positive window / negative sampling of loss.py:74-127 for one sample (boxes: [n,>=2] CPU)."""
        c = self.config
        rs = c["anchor_bbox_feature"]["reduced_scale"]
        span = c["positive_range"]
        half = int(span / 2)
        positives, regress, owner = [], [], []
        for box in boxes:
            members = []
            cx = int((float(box[0]) * self._xs + self._xo) / rs)
            cy = int((float(box[1]) * self._ys + self._yo) / rs)
            if 0 <= cx <= H - 1 and 0 <= cy <= W - 1:
                for dx in range(span):
                    for dy in range(span):
                        px, py = cx - half + dx, cy - half + dy
                        if px < 0 or px > H - 1 or py < 0 or py > W - 1:
                            continue
                        positives.append([px, py])
                        if self.regress_type == 0 or (px == cx and py == cy):
                            members.append(len(regress))
                            regress.append([px, py])
            owner.append(members)
        np.random.shuffle(positives)
        positives = positives[:c["pos_sample_threshold"]] if len(positives) > c["pos_sample_threshold"] else positives
        negatives = []
        while len(negatives) <= c["neg_sample_threshold"]:
            cand = [np.random.randint(H), np.random.randint(W)]
            if cand in positives:
                continue
            negatives.append(cand)
        return positives, negatives, regress, owner

    # ------------------------------------------------------------------ device-side terms
    @staticmethod
    def _ce(score2, pos, neg, dev):
        n = torch.tensor(neg, dtype=torch.long, device=dev)
        out = F.cross_entropy(score2[:, n[:, 0], n[:, 1]].t(), torch.zeros(len(neg), dtype=torch.long, device=dev))
        if len(pos) > 0:
            p = torch.tensor(pos, dtype=torch.long, device=dev)
            out = F.cross_entropy(score2[:, p[:, 0], p[:, 1]].t(), torch.ones(len(pos), dtype=torch.long, device=dev)) + out
        return out

    @staticmethod
    def _smooth_l1(box, pred, anc):
        """loss.py:144-165: encode the box against every anchor of the window, mean Smooth-L1."""
        N = anc.shape[0]
        ref = box[:7].view(1, 1, 7)
        diag = torch.sqrt(anc[:, :, 3:4] ** 2 + anc[:, :, 4:5] ** 2)
        d = ref[:, :, 6] - anc[:, :, 6]
        target = torch.cat(((ref[:, :, 0:2] - anc[:, :, 0:2]) / diag, (ref[:, :, 2:3] - anc[:, :, 2:3]) / anc[:, :, 5:6],
                            torch.log(ref[:, :, 3:6] / anc[:, :, 3:6]), torch.atan2(torch.sin(d), torch.cos(d)).unsqueeze(-1)), -1)
        return F.smooth_l1_loss(pred.reshape(N, 2, 7), target, reduction="none").sum() * (1.0 / (N * 14))

    def forward(self, reference_bboxes_batch, num_ref_bbox_batch, predicted_class_feature_batch, predicted_regress_feature_batch):
        cls, reg = predicted_class_feature_batch, predicted_regress_feature_batch
        dev = cls.device
        B = reference_bboxes_batch.shape[0]
        H, W = cls.shape[-2:]
        anc = self.anchor_set.to(dev)
        boxes_host = reference_bboxes_batch.detach().cpu()
        total = torch.zeros(1, device=dev)
        acc = torch.zeros(1, device=dev)
        for b in range(B):
            nb = int(num_ref_bbox_batch[b])
            pos, neg, regress, owner = self.assign(boxes_host[b, :nb], H, W)
            lc = self._ce(cls[b, 0:2], pos, neg, dev) + self._ce(cls[b, 2:4], pos, neg, dev)
            lr = torch.zeros(1, device=dev)
            if len(regress) > 0:
                rp = torch.tensor(regress, dtype=torch.long, device=dev)
                for k in range(nb):
                    if len(owner[k]) == 0:
                        continue
                    sel = rp[owner[k]]
                    pred = reg[b][:, sel[:, 0], sel[:, 1]].t()
                    a = anc[:, :, sel[:, 0], sel[:, 1]].permute(2, 0, 1)
                    lr = lr + self._smooth_l1(reference_bboxes_batch[b, k].to(dev), pred, a)
            total = lc + self.config["regress_loss_gain"] * lr
            acc = acc + total
        if self.reduction == "last":
            return total
        return acc if self.reduction == "sum" else acc / B
