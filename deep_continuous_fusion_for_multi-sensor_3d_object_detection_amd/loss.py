"""Drop-in counterpart of the reference's loss.py (LossTotal) -- the train-step harness row.

Same constructor and call surface: LossTotal(config)(bboxes [B,max,9], num_boxes [B],
cls [B,4,h,w], reg [B,14,h,w]) -> [1] loss tensor.  Target assignment is host-side Python
driven by numpy's global RNG exactly like loss.py:74-127 (so np.random.seed pins it); the
device half (gathers, cross-entropy, Smooth-L1 and their gradients) is ONE HIP launch for CUDA tensors
(csrc/loss.hip, dcf_loss_fwd_bwd -- SURVEY.md §8(f) row N1); CPU tensors (the host-logic tests) go through
the same arithmetic as a handful of torch ops.

`loss_sampling: device` moves the target assignment onto the device too (csrc/loss.hip, dcf_loss_sample_fwd_bwd: windows,
subset, negatives, terms and gradients in one launch, a stateless counter hash seeded by `loss_seed` and the call count
instead of numpy's generator) -- no host loop, no index lists over PCIe; `compat` (default) is the mode pinned to the reference.

Reference quirks are kept behind `loss_reduction: last` (default): cross-entropy on already
soft-maxed scores (loss.py:17-20,139), 129 negatives (:125), only the last sample of the
batch contributes (:71).  'sum' / 'mean' accumulate over the batch instead.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import AnchorBoundingBoxFeature


class _FusedLoss(torch.autograd.Function):
    """loss = dcf_loss_fwd_bwd(head outputs, staged targets); the same launch leaves dL/d(head outputs) in dense maps,
    which backward hands to autograd scaled by the incoming gradient.  With `base` = the [B,32,h,w] head tensor that
    cls / reg are views of, the gradient goes straight to it (no slice-backward kernels)."""

    @staticmethod
    def forward(ctx, base, cls, reg, anc, di, df, B, HW, gain, reduction):
        from . import _hip as H
        src = base if base is not None else cls
        loss = torch.zeros(1, dtype=torch.float32, device=src.device)
        if base is not None:
            g = torch.zeros_like(base)
            gb = g.stride(0)
            gcls, greg = g, g[:, 4:]
            ctx.split = False
        else:
            gcls, greg = torch.zeros_like(cls), torch.zeros_like(reg)
            g = (gcls, greg)
            gb = None
            ctx.split = True
        H.call("dcf_loss_fwd_bwd", cls, cls.stride(0), reg, reg.stride(0), anc, di, df, B, HW, float(gain), int(reduction), loss,
               gcls, gcls.stride(0), greg, greg.stride(0), H.stream_ptr())
        ctx.g = g
        return loss

    @staticmethod
    def backward(ctx, go):
        g = ctx.g
        if ctx.split:
            return None, g[0] * go, g[1] * go, None, None, None, None, None, None, None
        return g * go, None, None, None, None, None, None, None, None, None


class _FusedLossSample(torch.autograd.Function):
    """dcf_loss_sample_fwd_bwd: target assignment + loss + gradients in one launch (loss_sampling: device)."""

    @staticmethod
    def forward(ctx, base, cls, reg, anc, boxes, nbox, geo, seed, gain, reduction, outs):
        from . import _hip as H
        src = base if base is not None else cls
        B, _, Hh, W = cls.shape
        loss = torch.zeros(1, dtype=torch.float32, device=src.device)
        if base is not None:
            g = torch.zeros_like(base)
            gcls, greg = g, g[:, 4:]
            ctx.split = False
        else:
            gcls, greg = torch.zeros_like(cls), torch.zeros_like(reg)
            g = (gcls, greg)
            ctx.split = True
        xs, xo, ys, yo, rs, span, rtype, pos_cap, neg_count = geo
        H.call("dcf_loss_sample_fwd_bwd", cls, cls.stride(0), reg, reg.stride(0), anc, boxes, nbox, boxes.shape[1], boxes.shape[2], B, Hh, W,
               float(xs), float(xo), float(ys), float(yo), float(rs), int(span), int(rtype), int(pos_cap), int(neg_count), int(seed),
               float(gain), int(reduction), loss, gcls, gcls.stride(0), greg, greg.stride(0),
               None if outs is None else outs[0], None if outs is None else outs[1], None if outs is None else outs[2], H.stream_ptr())
        ctx.g = g
        return loss

    @staticmethod
    def backward(ctx, go):
        g = ctx.g
        if ctx.split:
            return (None, g[0] * go, g[1] * go) + (None,) * 8
        return (g * go,) + (None,) * 10


class LossTotal(nn.Module):
    def __init__(self, config):
        super(LossTotal, self).__init__()
        self.config = config
        self.regress_type = config["regress_type"]
        self.reduction = config.get("loss_reduction", "last")
        self.sampling = config.get("loss_sampling", "compat")
        if self.sampling not in ("compat", "device"):
            raise ValueError("loss_sampling must be compat or device (got %r)" % (self.sampling,))
        self.seed = int(config.get("loss_seed", 0))
        self.calls = 0                     # device sampling: the call count enters the hash, so every step draws fresh lists
                                           # (saved with the checkpoint: Train.save_checkpoint / load_checkpoint)
        self.last_samples = None           # device sampling with keep_samples: (pos [B,cap], neg [B,n], counts [B,2]) of the last call
        self.keep_samples = False
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
        centres = boxes[:, :2].tolist() if hasattr(boxes, "tolist") else [(float(b[0]), float(b[1])) for b in boxes]
        f32 = np.float32
        for bx, by in centres:
            members = []
            # the reference does this arithmetic on 0-dim fp32 tensors (loss.py:85-86): one fp32 rounding per operation, then
            # truncation -- in double precision a centre within fp32 rounding of a cell boundary lands one cell lower
            cx = int((f32(bx) * f32(self._xs) + f32(self._xo)) / f32(rs))
            cy = int((f32(by) * f32(self._ys) + f32(self._yo)) / f32(rs))
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
        # same draws and the same rejections as the reference's `[x, y] in list` test (loss.py:117-126), with a set lookup
        taken = set((p[0], p[1]) for p in positives)
        # ... and drawn in batches: randint([H, W], size=(m, 2)) consumes the legacy generator exactly like m pairs of
        # scalar calls (checked in tests/test_host_logic.py); a batch never holds more pairs than are still missing, so
        # no draw goes unused and the stream stays where the reference's loop leaves it
        negatives = []
        want = c["neg_sample_threshold"] + 1
        while len(negatives) < want:
            for x, y in np.random.randint([H, W], size=(want - len(negatives), 2)).tolist():
                if (x, y) not in taken:
                    negatives.append([x, y])
        return positives, negatives, regress, owner

    def assign_arrays(self, boxes, H, W):
        """assign() for the hot path: the same lists as flat integer codes (cell = px * W + py) -- (positive cells, negative cells,
        regression cells, box of each regression cell, weight of each regression cell) -- with the same consumption of numpy's
        legacy generator: RandomState.shuffle draws the same random_interval sequence for a 1-D array of n codes as for a list
        of n pairs (and takes its C fast path there), the negatives are drawn in the same batches.  tests/test_host_logic.py
        pins lists and generator state to assign().  boxes: float32 ndarray [n, >= 2]."""
        c = self.config
        rs, span, cap = c["anchor_bbox_feature"]["reduced_scale"], c["positive_range"], c["pos_sample_threshold"]
        half = int(span / 2)
        f32 = np.float32
        nb = len(boxes)
        # loss.py:85-86 on fp32 scalars: one rounding per operation, then truncation toward zero
        if nb:
            cxs = ((boxes[:, 0] * f32(self._xs) + f32(self._xo)) / f32(rs)).astype(np.int64).tolist()
            cys = ((boxes[:, 1] * f32(self._ys) + f32(self._yo)) / f32(rs)).astype(np.int64).tolist()
        else:
            cxs = cys = []
        cells, rows, row_box, row_w = [], [], [], []
        centre_only = self.regress_type != 0
        for k in range(nb):
            cx, cy = cxs[k], cys[k]
            if cx < 0 or cx > H - 1 or cy < 0 or cy > W - 1:
                continue
            x0, x1 = max(cx - half, 0), min(cx - half + span - 1, H - 1)
            y0, y1 = max(cy - half, 0), min(cy - half + span - 1, W - 1)
            win = [px * W + py for px in range(x0, x1 + 1) for py in range(y0, y1 + 1)]      # the reference's order: dx, then dy
            cells += win
            if centre_only:
                rows.append(cx * W + cy); row_box.append(k); row_w.append(1.0 / 14)
            else:
                rows += win
                row_box += [k] * len(win)
                row_w += [1.0 / (len(win) * 14)] * len(win)
        pos = np.array(cells, dtype=np.int64)
        np.random.shuffle(pos)
        pos = pos[:cap]
        taken = set(pos.tolist())
        want = c["neg_sample_threshold"] + 1
        neg = []
        while len(neg) < want:
            dr = np.random.randint([H, W], size=(want - len(neg), 2))
            for code in (dr[:, 0] * W + dr[:, 1]).tolist():
                if code not in taken:
                    neg.append(code)
        return pos, np.array(neg, dtype=np.int64), np.array(rows, dtype=np.int64), np.array(row_box, dtype=np.int64), np.array(row_w, dtype=np.float32)

    # ------------------------------------------------------------------ device-side terms
    def _stage_arrays(self, ints, floats, dev):
        """_stage for numpy arrays (the CUDA path): straight into pinned buffers.  A ring of three buffer pairs: the one being
        refilled was last used three steps ago, so the wait for its copies never blocks -- with a single pair the host stalled
        here every step until the GPU had reached the previous step's loss (the host enqueues a step about as fast as the GPU
        runs it, so that wait set the pace)."""
        ni, nf = max(ints.size, 1), max(floats.size, 1)
        ring = getattr(self, "_stage_ring", None)
        if ring is None or ring[0][0].numel() < ni or ring[0][1].numel() < nf:
            ring = []
            for _ in range(3):
                a, b = torch.empty(max(ni, 1 << 16), dtype=torch.long).pin_memory(), torch.empty(max(nf, 1 << 12), dtype=torch.float32).pin_memory()
                ring.append([a, b, None, a.numpy(), b.numpy()])
            self._stage_ring, self._stage_slot = ring, 0
        st = ring[self._stage_slot]
        self._stage_slot = (self._stage_slot + 1) % len(ring)
        if st[2] is not None:
            st[2].synchronize()
        st[3][:ints.size] = ints
        st[4][:floats.size] = floats
        di = st[0][:ni].to(dev, non_blocking=True)
        df = st[1][:nf].to(dev, non_blocking=True)
        st[2] = torch.cuda.Event()
        st[2].record()
        return di, df

    def _stage(self, ints, floats, dev):
        """One pinned staging buffer per kind: all index lists / boxes of the step go up in two async copies,
        so the host never waits for the device while it builds the loss."""
        if dev.type != "cuda":
            return torch.tensor(ints, dtype=torch.long), torch.tensor(floats, dtype=torch.float32)
        ni, nf = max(len(ints), 1), max(len(floats), 1)
        st = getattr(self, "_stage_buf", None)
        if st is None or st[0].numel() < ni or st[1].numel() < nf:
            st = [torch.empty(max(ni, 1 << 16), dtype=torch.long).pin_memory(), torch.empty(max(nf, 1 << 12), dtype=torch.float32).pin_memory(), None]
            self._stage_buf = st
        if st[2] is not None:
            st[2].synchronize()          # the previous step's copies have long completed
        st[0][:len(ints)] = torch.tensor(ints, dtype=torch.long)
        st[1][:len(floats)] = torch.tensor(floats, dtype=torch.float32)
        di = st[0][:ni].to(dev, non_blocking=True)
        df = st[1][:nf].to(dev, non_blocking=True)
        st[2] = torch.cuda.Event()
        st[2].record()
        return di, df

    def _forward_hip(self, cls, reg, anc, ints, floats, plan, B, H, W):
        """Device half as one launch: the per-sample table goes in front of the index lists (one staging copy)."""
        head = []
        for (o, npos, nneg, nrow, of, nb) in plan:
            head += [o + 6 * B, npos, nneg, nrow, of, nb]
        di, df = self._stage(head + ints, floats, cls.device)
        HW = H * W
        base, cls, reg = self._head_views(cls, reg, H, W)
        red = {"last": 0, "sum": 1, "mean": 2}[self.reduction]
        return _FusedLoss.apply(base, cls, reg, anc, di, df, B, HW, self.config["regress_loss_gain"], red)

    def _forward_hip_arrays(self, cls, reg, anc, boxes_host, nbox, B, H, W):
        """The CUDA path of the compat mode: numpy target assignment (assign_arrays) packed as dcf_loss_fwd_bwd wants it
        (per-sample table, then per sample: positive, negative, regression cells, box of each regression cell | weights, boxes)."""
        bh = boxes_host.numpy() if boxes_host.dtype == torch.float32 else boxes_host.float().numpy()
        head = np.empty((B, 6), np.int64)
        parts_i, parts_f = [], []
        o, of = 6 * B, 0
        for b in range(B):
            nb = int(nbox[b])
            pos, neg, rows, row_box, row_w = self.assign_arrays(bh[b, :nb], H, W)
            head[b] = (o, pos.size, neg.size, rows.size, of, nb)
            parts_i += [pos, neg, rows, row_box]
            bx = bh[b, :nb, :7].reshape(-1)
            parts_f += [row_w, bx]
            o += pos.size + neg.size + 2 * rows.size
            of += row_w.size + bx.size
        di, df = self._stage_arrays(np.concatenate([head.reshape(-1)] + parts_i), np.concatenate(parts_f) if parts_f else np.zeros(0, np.float32), cls.device)
        base, cls, reg = self._head_views(cls, reg, H, W)
        red = {"last": 0, "sum": 1, "mean": 2}[self.reduction]
        return _FusedLoss.apply(base, cls, reg, anc, di, df, B, H * W, self.config["regress_loss_gain"], red)

    @staticmethod
    def _head_views(cls, reg, H, W):
        """(base, cls, reg): base = the contiguous [B,>=18,h,w] head tensor cls / reg are views of (the gradient then goes straight
        to it), or None."""
        HW = H * W
        ok = lambda t, c: t.dtype == torch.float32 and t.stride(1) == HW and t.stride(2) == W and t.stride(3) == 1 and t.shape[1] == c
        if not ok(cls, 4):
            cls = cls.float().contiguous()
        if not ok(reg, 14):
            reg = reg.float().contiguous()
        base = cls._base
        if not (base is not None and reg._base is base and base.dim() == 4 and base.is_contiguous() and base.shape[1] >= 18
                and cls.data_ptr() == base.data_ptr() and reg.data_ptr() == base.data_ptr() + 4 * HW * 4 and base.requires_grad):
            base = None
        return base, cls, reg

    def _forward_device_sampling(self, boxes, nbox, cls, reg, anc, B, H, W):
        c = self.config
        dev = cls.device
        # labels: [B,max,9] fp32 and the counts, one small asynchronous copy each when they arrive on the host
        if not boxes.is_cuda:
            ring = getattr(self, "_box_ring", None)
            if ring is None or ring[0][0].shape != boxes.shape:
                ring = self._box_ring = [[torch.empty(boxes.shape, dtype=torch.float32).pin_memory(), None] for _ in range(3)]
                self._box_slot = 0
            st = ring[self._box_slot]            # last used three steps ago: the wait below never blocks
            self._box_slot = (self._box_slot + 1) % 3
            if st[1] is not None:
                st[1].synchronize()
            st[0].copy_(boxes)
            boxes = st[0].to(dev, non_blocking=True)
            st[1] = torch.cuda.Event()
            st[1].record()
        else:
            boxes = boxes.float().contiguous()
        nb = nbox.to(device=dev, dtype=torch.int32, non_blocking=True) if torch.is_tensor(nbox) else torch.tensor([int(v) for v in nbox], dtype=torch.int32, device=dev)
        base, cls, reg = self._head_views(cls, reg, H, W)
        geo = (self._xs, self._xo, self._ys, self._yo, c["anchor_bbox_feature"]["reduced_scale"], c["positive_range"], self.regress_type,
               c["pos_sample_threshold"], c["neg_sample_threshold"] + 1)
        outs = None
        if self.keep_samples:
            outs = (torch.empty((B, geo[7]), dtype=torch.int32, device=dev), torch.empty((B, geo[8]), dtype=torch.int32, device=dev),
                    torch.empty((B, 2), dtype=torch.int32, device=dev))
            self.last_samples = outs
        # rank 0 / a single process: (seed, calls) as before; other data-parallel ranks draw from streams of their own
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        seed = (self.seed * 0x9E3779B1 + self.calls + rank * 0x85EBCA77C2B2AE63) & 0xFFFFFFFFFFFFFFFF
        self.calls += 1
        red = {"last": 0, "sum": 1, "mean": 2}[self.reduction]
        return _FusedLossSample.apply(base, cls, reg, anc, boxes, nb, geo, seed, c["regress_loss_gain"], red, outs)

    def forward(self, reference_bboxes_batch, num_ref_bbox_batch, predicted_class_feature_batch, predicted_regress_feature_batch):
        cls, reg = predicted_class_feature_batch, predicted_regress_feature_batch
        dev = cls.device
        B = reference_bboxes_batch.shape[0]
        H, W = cls.shape[-2:]
        if getattr(self, "_anc_dev", None) is None or self._anc_dev.device != dev:
            self._anc_dev = self.anchor_set.to(dev).reshape(2, 7, H * W)
        anc = self._anc_dev
        if self.sampling == "device":
            if dev.type != "cuda":
                raise RuntimeError("loss_sampling: device needs CUDA tensors (the host path is loss_sampling: compat)")
            return self._forward_device_sampling(reference_bboxes_batch, num_ref_bbox_batch, cls, reg, anc, B, H, W)
        # pass CPU boxes (what a DataLoader yields) to avoid a device round trip
        boxes_host = reference_bboxes_batch.detach().cpu() if reference_bboxes_batch.is_cuda else reference_bboxes_batch.detach()
        if dev.type == "cuda":
            return self._forward_hip_arrays(cls, reg, anc, boxes_host, num_ref_bbox_batch, B, H, W)
        # ---- host: target assignment for every sample, packed into flat lists
        ints, floats, plan = [], [], []
        for b in range(B):
            nb = int(num_ref_bbox_batch[b])
            pos, neg, regress, owner = self.assign(boxes_host[b, :nb], H, W)
            rows, row_box, row_w = [], [], []
            for k in range(nb):
                for m in owner[k]:
                    rows.append(regress[m][0] * W + regress[m][1])
                    row_box.append(k)
                    row_w.append(1.0 / (len(owner[k]) * 14))
            o = len(ints)
            ints += [p[0] * W + p[1] for p in pos] + [q[0] * W + q[1] for q in neg] + rows + row_box
            of = len(floats)
            floats += row_w + boxes_host[b, :nb, :7].reshape(-1).tolist()
            plan.append((o, len(pos), len(neg), len(rows), of, nb))
        di, df = self._stage(ints, floats, dev)
        # ---- device: gathers + CE + Smooth-L1, vectorised per sample
        total = torch.zeros(1, device=dev)
        acc = torch.zeros(1, device=dev)
        for b in range(B):
            o, npos, nneg, nrow, of, nb = plan[b]
            pos_i, neg_i = di[o:o + npos], di[o + npos:o + npos + nneg]
            lc = torch.zeros(1, device=dev)
            for a in range(2):                                   # per anchor: loss.py:64-65
                sc = cls[b, 2 * a:2 * a + 2].reshape(2, H * W)
                term = F.cross_entropy(sc[:, neg_i].t(), torch.zeros(nneg, dtype=torch.long, device=dev))
                if npos > 0:
                    term = F.cross_entropy(sc[:, pos_i].t(), torch.ones(npos, dtype=torch.long, device=dev)) + term
                lc = lc + term
            lr = torch.zeros(1, device=dev)
            if nrow > 0:
                rows = di[o + npos + nneg:o + npos + nneg + nrow]
                rbox = di[o + npos + nneg + nrow:o + npos + nneg + 2 * nrow]
                w_row = df[of:of + nrow]
                boxes = df[of + nrow:of + nrow + nb * 7].reshape(nb, 7)
                pred = reg[b].reshape(14, H * W)[:, rows].t().reshape(nrow, 2, 7)
                an = anc[:, :, rows].permute(2, 0, 1)            # [nrow,2,7]
                ref = boxes[rbox].unsqueeze(1)                   # [nrow,1,7]
                diag = torch.sqrt(an[:, :, 3:4] ** 2 + an[:, :, 4:5] ** 2)
                d = ref[:, :, 6] - an[:, :, 6]
                target = torch.cat(((ref[:, :, 0:2] - an[:, :, 0:2]) / diag, (ref[:, :, 2:3] - an[:, :, 2:3]) / an[:, :, 5:6],
                                    torch.log(ref[:, :, 3:6] / an[:, :, 3:6]), torch.atan2(torch.sin(d), torch.cos(d)).unsqueeze(-1)), -1)
                per_row = F.smooth_l1_loss(pred, target, reduction="none").sum((1, 2))
                lr = lr + (per_row * w_row).sum()                # = sum over boxes of the per-box mean (loss.py:163,186)
            total = lc + self.config["regress_loss_gain"] * lr
            acc = acc + total
        if self.reduction == "last":
            return total
        return acc if self.reduction == "sum" else acc / B
