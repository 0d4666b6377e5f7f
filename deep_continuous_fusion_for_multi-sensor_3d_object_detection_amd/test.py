"""Drop-in counterpart of the reference's test.py `Test` class (the part train.py uses).

Test(net, config) puts the module in eval mode exactly like test.py:37 (which is why the
reference trains with eval-mode BatchNorm) and evaluates loss + score-threshold
counts for a batch.  The rotated-NMS / precision-recall post-processing (test.py:88-206; SURVEY.md
section 8(f) N2) keeps the reference's behaviour -- greedy suppression in INPUT order (no score sort), IoU
candidates shifted by 1e-4, touching rectangles suppress each other in the SAT flavour, and the aliased
per-box TP history (test.py:204 appends the same dict object every time) -- pinned by tests/golden/eval.npz.
Boxes that live on the GPU take the HIP kernels of csrc/evalpost.hip (score filter + compaction, pairwise overlap
matrix + one-wave greedy scan, bird's-eye-IoU matching); boxes handed over as host tensors take the numpy statement
of the same definitions (evalgeom.py), which is also what the device kernels are checked against.
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

    @staticmethod
    def _on_device(pred_bboxes):
        from .ops import EVAL_NMS_CAP
        ok = [isinstance(b, torch.Tensor) and b.is_cuda and b.dim() == 2 and b.shape[0] <= EVAL_NMS_CAP for b in pred_bboxes]
        return bool(ok) and all(ok)

    @staticmethod
    def _nms_device(pred_bboxes, mode, thr=0.01):
        from . import ops
        flags = [ops.eval_nms(b, mode, thr)[0] if b.shape[0] else None for b in pred_bboxes]
        out = []
        for b, f in zip(pred_bboxes, flags):
            idx = [] if f is None else torch.nonzero(f.cpu()).flatten().tolist()      # the one device -> host transfer
            out.append([b[i] for i in idx])
        return out

    def NMS_IOU(self, pred_bboxes, nms_iou_score_theshold=0.01):
        """test.py:110-140: 3-D IoU above the threshold suppresses; the survivor's centre is nudged by 1e-4 as there."""
        if self._on_device(pred_bboxes):
            return self._nms_device(pred_bboxes, "iou", nms_iou_score_theshold)

        def over(c, k):
            cc = EG.box_corners(c[:3], c[3:6], c[6])
            kc = EG.box_corners(k[:3] + 0.0001, k[3:6], k[6])
            return EG.rotated_iou(cc, kc)[0] > nms_iou_score_theshold
        return self._nms(pred_bboxes, over)

    def NMS_SAT(self, pred_bboxes):
        """test.py:142-175: any overlap (or contact) of the bird's-eye rectangles suppresses."""
        if self._on_device(pred_bboxes):
            return self._nms_device(pred_bboxes, "sat")

        def over(c, k):
            return EG.rects_overlap(EG.bev_rect(c[:2], c[3:5], c[6]), EG.bev_rect(k[:2], k[3:5], k[6]))
        return self._nms(pred_bboxes, over)

    def precision_recall_singleshot(self, pred_bboxes, ref_bboxes):
        """test.py:177-206: a predicted box is a true positive at threshold t if its bird's-eye IoU with any labelled
        box (last column == 1) exceeds t."""
        dev = [p for p in pred_bboxes if p is not None and len(p) and isinstance(p[0], torch.Tensor) and p[0].is_cuda]
        if dev:
            return self._precision_recall_device(pred_bboxes, ref_bboxes)
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

    def _precision_recall_device(self, pred_bboxes, ref_bboxes):
        """The same counters from dcf_eval_match: one launch per sample, ten integers back at the end."""
        from . import ops
        device = next(p[0].device for p in pred_bboxes if p is not None and len(p))
        thr = torch.tensor(self.IOU_threshold, dtype=torch.float64, device=device)
        tp = torch.zeros(len(self.IOU_threshold), dtype=torch.int32, device=device)
        refs_dev = ref_bboxes.to(device=device, dtype=torch.float32)
        for b in range(ref_bboxes.shape[0]):
            kept = pred_bboxes[b]
            if kept is not None and len(kept):
                ops.eval_match(torch.stack([k.float() for k in kept], 0), refs_dev[b], thr, tp)
                self.num_P += len(kept)
                self.num_TP_set_per_predbox.extend([self.num_TP_set] * len(kept))   # same object every time, as in the reference
            self.num_T += int((ref_bboxes[b][:, -1] == 1).sum())
        for t, v in zip(self.IOU_threshold, tp.cpu().tolist()):
            self.num_TP_set[t] += v

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

    def get_bboxes_device(self, pred, score_threshold=None):
        """test.py:88-108 as one launch (dcf_eval_score_filter) on the model output pred [B,32,h,w]: [n,7] boxes per sample,
        anchor 0's in raster order, then anchor 1's.  Falls back to get_bboxes when a sample has more than 4096 candidates."""
        from . import ops
        thr = self.config["score_threshold"] if score_threshold is None else score_threshold
        boxes, count = ops.eval_score_filter(pred, thr)
        n = count.cpu().tolist()
        if max(n) > boxes.shape[1]:
            _, _, pb = torch.split(pred, [4, 14, 14], dim=1)
            return self.get_bboxes(pred[:, 0:4], pb, thr)
        return [boxes[b, :n[b]] for b in range(pred.shape[0])]

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
            boxes = self.get_bboxes_device(pred) if pred.is_cuda else self.get_bboxes(pred_cls, pred_bbox)
        # test.py:84-86: SAT suppression, then the precision / recall counters (on the device when the boxes are there and a
        # sample has at most 4096 candidates; the numpy statement otherwise)
        if not self._on_device(boxes):
            boxes_nms = [b.float().cpu() for b in boxes]
        else:
            boxes_nms = boxes
        self.refined_bbox = self.NMS_SAT(boxes_nms)
        self.precision_recall_singleshot(self.refined_bbox, object_data.detach().float().cpu())
        return self.loss_value.item(), boxes
