"""Drop-in counterpart of the reference's test.py `Test` class (the part train.py uses).

Test(net, config) puts the module in eval mode exactly like test.py:37 (which is why the
reference trains with eval-mode BatchNorm) and evaluates loss + score-threshold
counts for a batch.  The rotated-NMS / AP post-processing (test.py:88-250, IOU.py,
separation_axis_theorem.py) is host-side evaluation code outside the train-step hot path
 and is not rebuilt here.
"""
import torch
import torch.nn as nn

from .loss import LossTotal


class Test(nn.Module):
    def __init__(self, pre_trained_net, config):
        super(Test, self).__init__()
        self.net = pre_trained_net
        self.net.eval()
        self.config = config
        self.loss_total = LossTotal(config)
        self.initialize_ap()

    def initialize_ap(self):
        self.num_T = 0
        self.num_P = 0
        self.loss_value = None

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
        self.num_T += int(sum(int(n) for n in num_ref_box))
        self.num_P += int(sum(b.shape[0] for b in boxes))
        return self.loss_value.item(), boxes
