"""Drop-in counterpart of the reference's test.py `Test` class (the part train.py uses).

Test(net, config) puts the module in eval mode exactly like test.py:37 (which is why the
reference trains with eval-mode BatchNorm) and evaluates loss + score-threshold
counts for a batch.  The rotated-NMS / precision-recall post-processing (test.py:110-250; SURVEY.md
section 8(f) N2) is host-side evaluation code outside the train-step hot path: restated here on numpy
(evalgeom.py) with the reference's behaviour -- greedy suppression in INPUT order (no score sort), IoU
candidates shifted by 1e-4, touching rectangles suppress each other in the SAT flavour, and the aliased
per-box TP history (test.py:204 appends the same dict object every time) -- pinned by tests/golden/eval.npz.
"""
import numpy as np
import torch
import torch.nn as nn

from . import evalgeom as EG
from .loss import LossTotal


class Test(nn.Module):
    def __init__(self, pre_trained_net, config):
        super(Test, self).__init__()
        self.net = pre_trained_net
        self.net.eval()
        self.config = config
        self.loss_total = LossTotal(config)
        self.initialize_ap()

    IOU_threshold = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]

    def initialize_ap(self):
        self.num_T = 0
        self.num_P = 0
        self.num_TP_set = {t: 0 for t in self.IOU_threshold}
        self.num_TP_set_per_predbox = []
        self.loss_value = None

    def get_num_TP_set(self):
        return self.num_TP_set

    # ------------------------------------------------------------------ post-processing (host)
    @staticmethod
    def _np(box):
        return box.detach().cpu().numpy().astype(np.float64) if isinstance(box, torch.Tensor) else np.asarray(box, dtype=np.float64)

    def _nms(self, pred_bboxes, overlaps):
        """Greedy suppression in input order (test.py:110-175): a box survives iff it overlaps none of the survivors."""
        out = []
        for boxes in pred_bboxes:
            kept, kept_np = [], []
            for i in range(len(boxes)):
                cand = self._np(boxes[i])
                if all(not overlaps(cand, k) for k in kept_np):
                    kept.append(boxes[i])
                    kept_np.append(cand)
            out.append(kept)
        return out

    def NMS_IOU(self, pred_bboxes, nms_iou_score_theshold=0.01):
        """test.py:110-140: 3-D IoU above the threshold suppresses; the survivor's centre is nudged by 1e-4 as there."""
        def over(c, k):
            cc = EG.box_corners(c[:3], c[3:6], c[6])
            kc = EG.box_corners(k[:3] + 0.0001, k[3:6], k[6])
            return EG.rotated_iou(cc, kc)[0] > nms_iou_score_theshold
        return self._nms(pred_bboxes, over)

    def NMS_SAT(self, pred_bboxes):
        """test.py:142-175: any overlap (or contact) of the bird's-eye rectangles suppresses."""
        def over(c, k):
            return EG.rects_overlap(EG.bev_rect(c[:2], c[3:5], c[6]), EG.bev_rect(k[:2], k[3:5], k[6]))
        return self._nms(pred_bboxes, over)

    def precision_recall_singleshot(self, pred_bboxes, ref_bboxes):
        """test.py:177-206: a predicted box is a true positive at threshold t if its bird's-eye IoU with any labelled
        box (last column == 1) exceeds t."""
        for b in range(ref_bboxes.shape[0]):
            refs = [self._np(r) for r in ref_bboxes[b] if float(r[-1]) == 1]
            ref_c = [EG.box_corners(r[:3], r[3:6], r[6]) for r in refs]
            if pred_bboxes[b] is not None:
                for pb in pred_bboxes[b]:
                    self.num_P += 1
                    p = self._np(pb)
                    pc = EG.box_corners(p[:3], p[3:6], p[6])
                    hit = set()
                    for rc in ref_c:
                        iou2d = EG.rotated_iou(pc, rc)[1]
                        hit.update(t for t in self.IOU_threshold if iou2d > t)
                    for t in hit:
                        self.num_TP_set[t] += 1
                    self.num_TP_set_per_predbox.append(self.num_TP_set)      # same object every time, as in the reference
            self.num_T += len(refs)

    def display_average_precision(self, plot_AP_graph=False):
        """test.py:208-242 without the matplotlib file output: (precision, recall) curves per IoU threshold."""
        precisions = {t: [1] for t in self.IOU_threshold}
        recalls = {t: [0] for t in self.IOU_threshold}
        for n, tp in enumerate(self.num_TP_set_per_predbox, 1):
            for t in self.IOU_threshold:
                precisions[t].append(tp[t] / n)
                recalls[t].append(tp[t] / self.num_T)
        self.total_precision = {t: self.num_TP_set[t] / (self.num_P + 0.01) for t in self.IOU_threshold}
        self.total_recall = {t: self.num_TP_set[t] / (self.num_T + 0.01) for t in self.IOU_threshold}
        return precisions, recalls

    def get_num_T(self):
        return self.num_T

    def get_num_P(self):
        return self.num_P

    def get_bboxes(self, pred_cls, pred_bbox, score_threshold=None):
        """test.py:88-108: anchors whose positive-class score exceeds the threshold -> [n,7] boxes per sample."""
        thr = self.config["score_threshold"] if score_threshold is None else score_threshold
        out = []
        for b in range(pred_cls.shape[0]):
            boxes = []
            for a in range(2):
                mask = pred_cls[b, 2 * a + 1] > thr
                boxes.append(pred_bbox[b, 7 * a:7 * a + 7][:, mask].t())
            out.append(torch.cat(boxes, 0))
        return out

    def get_eval_value_onestep(self, lidar_voxel, camera_image, object_data, num_ref_box, **extra):
        with torch.no_grad():
            pred = self.net(lidar_voxel, camera_image, **extra)
            pred_cls, pred_reg, pred_bbox = torch.split(pred, [4, 14, 14], dim=1)
            self.loss_value = self.loss_total(object_data.to(pred.device), num_ref_box, pred_cls, pred_reg)
            boxes = self.get_bboxes(pred_cls, pred_bbox)
        # test.py:84-86: SAT suppression, then the precision / recall counters
        self.refined_bbox = self.NMS_SAT([b.float().cpu() for b in boxes])
        self.precision_recall_singleshot(self.refined_bbox, object_data.detach().float().cpu())
        return self.loss_value.item(), boxes
