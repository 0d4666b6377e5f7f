"""Drop-in module surface of the reference's model.py on the MI355X engine.

Same class names, constructor arguments, forward signatures and state_dict keys as
/root/reference/model.py; underneath, every tensor op is a HIP kernel reached through the
C ABI (engine.Plan + backend_hip.HipBackend).  There is no torch/CPU compute fallback:
calling forward on CPU tensors raises.

  ObjectDetection_DCF(config)(x_lidar [B,Cz,L,W] f32, x_image [B,3,H,W] u8) -> [B,32,L/4,W/4] f32
      = cat(cls[4], reg[14], bbox[14])                       (model.py:176-204)
  optional extra arguments (points, uv, n_valid) feed the continuous-fusion layers the
  reference leaves as a TODO (model.py:199-203); without them -- or with config
  fusion.enabled = False -- the output is the reference's LiDAR-only forward.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _hip as H
from .engine import ParamTable, Plan, StackPlan


class AnchorBoundingBoxFeature(nn.Module):
    """model.py:82-113: dense anchor tensor [14,h,w] (2 anchors x (x,y,z,l,w,h,yaw)).

    Built on the host with the same torch.linspace calls (endpoints inclusive), so it is
    bit-identical to the reference; the engine uploads it once per device instead of once
    per forward (model.py:125)."""

    def __init__(self, config):
        super(AnchorBoundingBoxFeature, self).__init__()
        self.config = config
        a = config["anchor_bbox_feature"]
        self.f_height = int(config["voxel_length"] / a["reduced_scale"])
        self.f_width = int(config["voxel_width"] / a["reduced_scale"])
        self.width, self.length, self.height = a["width"], a["length"], a["height"]

    def forward(self):
        h, w, c = self.f_height, self.f_width, self.config
        xs = torch.linspace(c["lidar_x_min"], c["lidar_x_max"], h).view(h, 1).expand(h, w)
        ys = torch.linspace(c["lidar_y_min"], c["lidar_y_max"], w).view(1, w).expand(h, w)
        one = torch.ones(h, w)
        common = [xs, ys, one * (-4.5), one * self.length, one * self.width, one * self.height]
        return torch.stack(common + [one * 0] + common + [one * 3.1415926 / 2], 0).contiguous()


class _Leaf(nn.Module):
    """Container node so that parameters carry the reference's dotted state_dict names."""


class _RunPlan(torch.autograd.Function):
    """Bridges the hand-written forward/backward of engine.Plan into torch autograd so that
    `loss.backward()` (train.py:35) drives it.  Parameter gradients are written straight
    into the flat gradient arena (each Parameter's .grad is a view of it)."""

    @staticmethod
    def forward(ctx, token, model, x_lidar, x_image, geom, need):
        ctx.model = model
        pred = model._plan.forward(model._backend, x_lidar, x_image, geom, save=need)
        ctx.saved_graph = need
        model._fwd_serial = ctx.serial = getattr(model, "_fwd_serial", 0) + 1
        return pred

    @staticmethod
    def backward(ctx, gpred):
        model = ctx.model
        if not ctx.saved_graph:
            raise RuntimeError("backward through a forward that ran without saving activations")
        if ctx.serial != model._fwd_serial:
            # the plan keeps ONE set of saved activations (and one gradient arena that every backward overwrites): unlike an
            # nn.Module tree under autograd, forward(a); forward(b); loss_a.backward() cannot work -- say so instead of silently
            # differentiating b's activations
            raise RuntimeError("backward of a stale forward: this module keeps the activations of its LAST forward only "
                               "(call backward before the next forward; gradients are overwritten, not accumulated)")
        model._plan.backward(model._backend, gpred.contiguous())
        model._bind_grads()
        return None, None, None, None, None, None


class _RunGraphs(torch.autograd.Function):
    """Same bridge as _RunPlan, but the ~120 forward and ~250 backward launches are two captured HIP graphs
    replayed with one call each (config['hip_graphs']): launch-bound gaps between the many small kernels and
    the per-launch host work disappear.  Inputs are copied into the graphs' static buffers first."""

    @staticmethod
    def forward(ctx, token, model, x_lidar, x_image, geom):
        ctx.model = model
        model._fwd_serial = ctx.serial = getattr(model, "_fwd_serial", 0) + 1
        return model._graphs.run_forward(x_lidar, x_image, geom)

    @staticmethod
    def backward(ctx, gpred):
        model = ctx.model
        if ctx.serial != model._fwd_serial:
            raise RuntimeError("backward of a stale forward: this module keeps the activations of its LAST forward only")
        model._graphs.run_backward(gpred)
        model._bind_grads()
        return None, None, None, None, None


class _GraphSet(object):
    """Static buffers and the captured graphs of one input signature (tensor shapes + rows of the per-point fusion
    tensors): g_img = weight preparation + camera stream, g_lid = LiDAR stream with the fusion sites and the heads,
    g_bwd = the whole backward.  The three share one memory pool (activations saved by the forward graphs are read by
    the backward graph).  When the geometry arrives in the trainer's persistent buffer sets (geom["static"]), the graphs
    are captured ON those buffers: a replay then needs no staging copy of the voxel image, the points or the KNN maps."""

    def __init__(self, model, x_lidar, x_image, geom, n_rows):
        m, K = model, model._backend
        self.static = bool(geom is not None and geom.get("static")) or (geom is None)
        self.sx = x_lidar if (geom is not None and geom.get("static")) else x_lidar.clone()
        self.simg = None if x_image is None else x_image.clone()
        self.sgeom = None
        if geom is not None and geom.get("xyz") is not None:
            keep = (lambda t: t) if geom.get("static") else (lambda t: t.clone())
            self.sgeom = dict(xyz=keep(geom["xyz"]), uv=keep(geom["uv"]), cnt=keep(geom["cnt"]),
                              idx=[keep(t) for t in geom["idx"]], aff=geom["aff"], n_rows=n_rows)
            if geom.get("inv") is not None:
                self.sgeom["inv"] = tuple(keep(t) for t in geom["inv"])
                self.sgeom["inv_nmax"] = geom["inv_nmax"]
        self.copy_geom = self.sgeom is not None and not geom.get("static")
        self.copy_x = not (geom is not None and geom.get("static"))
        split = m._plan.with_image and self.sgeom is not None
        # eager warm-up step on the static buffers: lazy allocations (slabs, anchors, workspaces) happen here.  (Train-mode
        # BatchNorm: the warm-up is not a step of the run -- the running statistics it moved are put back.)
        keep_stats = m._bufflat.clone() if K.bn_train else None
        K.prepare()
        pred = m._plan.forward(K, self.sx, self.simg, self.sgeom, save=True)
        m._plan.backward(K, torch.zeros_like(pred))
        if keep_stats is not None:
            m._bufflat.copy_(keep_stats)
        torch.cuda.synchronize()
        # the captured launches carry raw addresses of the backend's per-signature arenas (split-K slabs, dbeta sums, layer
        # table) and of its weight images: this set keeps them alive for as long as its graphs can be replayed, whatever the
        # backend's own cache evicts meanwhile
        self._arenas = (K.slabs, K.gsum, K.table, K.warena, K.ssarena)
        self.g_img = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_img):
            K.prepare()
            self.sfmap = m._plan.forward_image(K, self.simg, save=True) if split else None
        # the LiDAR stream in two graphs: what only needs the voxel image (layer1, layer2's blocks) ...
        self.g_lid_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_lid_a, pool=self.g_img.pool()):
            self.sxa = m._plan.forward(K, self.sx, self.simg, self.sgeom, save=True, fmap=self.sfmap, phase=1)
        # ... and everything from the first fusion site on, which needs the KNN maps of the geometry side stream
        self.g_lid = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_lid, pool=self.g_img.pool()):
            self.spred = m._plan.forward(K, self.sx, self.simg, self.sgeom, save=True, fmap=self.sfmap, phase=2, resume=self.sxa)
        self.sgpred = torch.zeros_like(self.spred)
        self.g_bwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_bwd, pool=self.g_img.pool()):
            m._plan.backward(K, self.sgpred)


class _StepGraphs(object):
    """Captured-graph execution of the train step (config['hip_graphs']).  The geometry stays outside the graphs, on the
    caller's side stream (train.Train.geometry_async): the camera-stream graph is replayed first and overlaps it, the
    LiDAR-stream graph follows once the voxel grid and the KNN maps are there.  One set of graphs per input signature;
    the signature includes the row count of the per-point fusion tensors -- the valid-point count rounded up to ROWS_STEP, a
    coarse step so that real frames (whose in-frustum counts vary by a few per cent) share one or two sets -- and the
    address of the geometry buffers (the trainer's two buffer sets: two graph sets per signature).  Sets are evicted
    least-recently-used; a run that keeps meeting NEW shapes / row buckets (more than MAX_MISSES of them) falls back to eager
    launches for good instead of capturing for ever."""
    MAX_SETS = 16          # (the trainer's two geometry buffer sets double every shape / row-count bucket)
    ROWS_STEP = 8192
    MAX_MISSES = 24        # distinct (shapes, row bucket) signatures, whatever buffer set they arrived in

    def __init__(self, model):
        import collections
        self.model = model
        self.sets = collections.OrderedDict()
        self.cur = None
        self.misses = 0
        self.seen = set()
        self.disabled = False

    @classmethod
    def _rows(cls, geom):
        if geom is None or geom.get("xyz") is None:
            return 0
        n = geom["xyz"].shape[1]
        ch = geom.get("cnt_host")
        if ch is not None:
            geom["cnt_event"].synchronize()
            n = min(n, max(cls.ROWS_STEP, (int(ch.max()) + cls.ROWS_STEP - 1) // cls.ROWS_STEP * cls.ROWS_STEP))
        return n

    def run_forward(self, x_lidar, x_image, geom):
        cur = torch.cuda.current_stream()
        n_rows = self._rows(geom)
        has_geo = geom is not None and geom.get("xyz") is not None
        sig = (tuple(x_lidar.shape), x_lidar.dtype, None if x_image is None else tuple(x_image.shape),
               None if not has_geo else (tuple(geom["xyz"].shape), tuple(geom["idx"][0].shape), geom.get("inv") is not None), n_rows,
               x_lidar.data_ptr() if (geom is not None and geom.get("static")) else 0)
        st = self.sets.get(sig)
        if st is None:
            if geom is not None:
                for k in ("voxel_event", "event", "inv_event"):
                    if geom.get(k) is not None:
                        cur.wait_event(geom[k])
            while len(self.sets) >= self.MAX_SETS:
                self.sets.popitem(last=False)
            if sig[:-1] not in self.seen:          # the other buffer set of a known signature is not a new kind of input
                self.seen.add(sig[:-1])
                self.misses += 1
            st = self.sets[sig] = _GraphSet(self.model, x_lidar, x_image, geom if has_geo or geom is None else None, n_rows)
        else:
            self.sets.move_to_end(sig)
        self.cur = (st, geom)
        if st.simg is not None:
            st.simg.copy_(x_image)
        st.g_img.replay()                              # weight images + camera stream: overlaps the geometry side stream
        if geom is not None and geom.get("voxel_event") is not None:
            cur.wait_event(geom["voxel_event"])
        if st.copy_x:
            st.sx.copy_(x_lidar)
        st.g_lid_a.replay()                            # layer1 + layer2's blocks: the KNN may still be running
        if st.sgeom is not None:
            if geom.get("event") is not None:
                cur.wait_event(geom["event"])
            if st.copy_geom:
                st.sgeom["xyz"].copy_(geom["xyz"]); st.sgeom["uv"].copy_(geom["uv"]); st.sgeom["cnt"].copy_(geom["cnt"])
                for d, s_ in zip(st.sgeom["idx"], geom["idx"]):
                    d.copy_(s_)
        st.g_lid.replay()
        return st.spred

    def run_backward(self, gpred):
        st, geom = self.cur
        if st.sgeom is not None and st.sgeom.get("inv") is not None:
            if geom.get("inv_event") is not None:
                torch.cuda.current_stream().wait_event(geom["inv_event"])
            if st.copy_geom:
                for d, s_ in zip(st.sgeom["inv"], geom["inv"]):
                    d.copy_(s_)
        st.sgpred.copy_(gpred)
        st.g_bwd.replay()


class _FlatParamModule(nn.Module):
    """nn.Module whose Parameters are strided views of ONE flat fp32 arena laid out by an engine.ParamTable (gradients:
    a second arena with the same offsets), registered under the reference's dotted state_dict names."""

    @property
    def _table(self):
        return self._plan.table

    def _on_moved(self):
        """Device changed: drop everything that holds device addresses."""
        self._backend = None

    # ------------------------------------------------------------------ parameters
    def _build_parameters(self, device="cpu"):
        t = self._table
        self._flat = torch.zeros(max(t.n_params, 4), dtype=torch.float32, device=device)
        self._gradflat = torch.zeros_like(self._flat)
        self._bufflat = torch.zeros(max(t.n_buffers, 4), dtype=torch.float32, device=device)
        self._param_list, self._param_meta = [], []
        for key, shape, off, n, layout in t.entries:
            p = nn.Parameter(ParamTable.view(self._flat, shape, off, n, layout))
            self._register(key, p, False)
            self._param_list.append(p)
            self._param_meta.append((shape, off, n, layout))
        self._buf_meta = []
        self._nbt_keys = [key for key, shape, off, n in t.buffers if key.endswith("num_batches_tracked")]
        # every BatchNorm's num_batches_tracked is a 0-d view of ONE int64 tensor: a train-mode forward bumps them all with one
        # launch (62 launches per cfg2 step as separate tensors)
        self._nbtflat = torch.zeros((max(len(self._nbt_keys), 1),), dtype=torch.long, device=device)
        for key, shape, off, n in t.buffers:
            if key.endswith("num_batches_tracked"):
                self._register(key, self._nbtflat[self._nbt_keys.index(key)], True)
            else:
                self._register(key, self._bufflat[off:off + n].view(shape), True)
                self._buf_meta.append((key, shape, off, n))

    def _register(self, key, tensor, is_buffer):
        parts = key.split(".")
        node = self
        for name in parts[:-1]:
            if name not in node._modules:
                node.add_module(name, _Leaf())
            node = node._modules[name]
        if is_buffer:
            node.register_buffer(parts[-1], tensor)
        else:
            node.register_parameter(parts[-1], tensor)

    def _resolve(self, key):
        parts = key.split(".")
        node = self
        for name in parts[:-1]:
            node = node._modules[name]
        return node, parts[-1]

    def reset_parameters(self, zero_init_last=False):
        """nn.Conv2d / nn.Linear / nn.BatchNorm2d default initialisation (what the reference gets)."""
        with torch.no_grad():
            self._flat.zero_()
            for p, (shape, off, n, layout) in zip(self._param_list, self._param_meta):
                if len(shape) >= 2:
                    bound = 1.0 / math.sqrt(float(np.prod(shape[1:])))
                    p.uniform_(-bound, bound)
            for (key, shape, off, n, layout), p in zip(self._table.entries, self._param_list):
                leaf = key.split(".")[-1]
                if len(shape) == 1 and leaf == "weight":
                    p.fill_(1.0)
                elif len(shape) == 1 and "fusion" in key:      # linear biases
                    p.uniform_(-0.05, 0.05)
                if zero_init_last and "fusion" in key and ".fc2." in key:
                    p.zero_()
            for key, shape, off, n in self._buf_meta:
                self._bufflat[off:off + n].fill_(1.0 if key.endswith("running_var") else 0.0)

    def _apply(self, fn, recurse=True):
        """.cuda()/.to(): move the arenas as a whole and re-point every Parameter at the new arena."""
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("master parameters stay fp32; choose the compute dtype with config['dtype']")
        if new_flat.device == self._flat.device:
            return self
        self._flat = new_flat.contiguous()
        self._gradflat = torch.zeros_like(self._flat)
        self._bufflat = fn(self._bufflat).contiguous()
        for p, (shape, off, n, layout) in zip(self._param_list, self._param_meta):
            p.data = ParamTable.view(self._flat, shape, off, n, layout)
            p.grad = None
        for key, shape, off, n in self._buf_meta:
            node, leaf = self._resolve(key)
            node._buffers[leaf] = self._bufflat[off:off + n].view(shape)
        self._nbtflat = fn(self._nbtflat).contiguous()
        for i, key in enumerate(self._nbt_keys):
            node, leaf = self._resolve(key)
            node._buffers[leaf] = self._nbtflat[i]
        self._on_moved()
        return self

    @property
    def _nbt(self):
        out = []
        for key in self._nbt_keys:
            node, leaf = self._resolve(key)
            out.append(node._buffers[leaf])
        return out

    def _bind_grads(self):
        for p, (shape, off, n, layout) in zip(self._param_list, self._param_meta):
            p.grad = ParamTable.view(self._gradflat, shape, off, n, layout)

    # flat views for a fused optimiser / gradient all-reduce (train.py)
    @property
    def flat_params(self):
        return self._flat

    @property
    def flat_grads(self):
        return self._gradflat


class ObjectDetection_DCF(_FlatParamModule):
    def __init__(self, config):
        super(ObjectDetection_DCF, self).__init__()
        self.config = config
        fu = dict(config.get("fusion") or {})
        self.fusion_enabled = bool(fu.get("enabled", False))
        self.K = int(fu.get("K", 3))
        self.r_max = fu.get("r_max", None)
        self.cf = int(fu.get("image_channels", 64))
        dt = config.get("dtype", "f32")
        # "fp8": forward convolutions with Cin >= fp8_min_cin take e4m3 operands (csrc/conv_fp8.hip); storage, the
        # backward and everything else are bf16 (BASELINE.json configs[4])
        self.fp8 = dt in ("fp8", "f8", "e4m3")
        self.dtype = H.dtype_code("bf16" if self.fp8 else dt)
        # "eval": running statistics always -- what train.py effectively does (test.py:37 puts the trained module in
        # eval mode before the first step); "train": batch statistics always; "module": follow
        # nn.Module.training exactly like nn.BatchNorm2d would.
        self.bn_mode = config.get("bn_mode", "eval")
        if self.bn_mode not in ("eval", "train", "module"):
            raise ValueError("bn_mode must be eval, train or module (got %r)" % (self.bn_mode,))
        stream = fu.get("image_stream", "resnet18")
        from .engine import IMAGE_ARCHS
        if self.fusion_enabled and stream not in IMAGE_ARCHS:
            raise NotImplementedError("image_stream=%r (one of %s)" % (stream, sorted(IMAGE_ARCHS)))
        # hip_graphs: True / False (default), or "auto" (bench.py's default) = replay captured graphs for batches of ONE frame in a single-rank run -- the
        # one configuration where eager launches can leave the GPU waiting for the host (round 5, two boxes: 254.9 frames/s replayed
        # against 250.4 eager, and 250.4 against 216.6 on a slower host: profiles/r05a_* / r05b_*; under replay the compute queue is
        # 96 % busy, profiles/r05f_timeline_b1.txt); at batch 2 the step is kernel-bound and eager launches keep the gradient buckets' overlap
        hg = config.get("hip_graphs", False)
        self.use_graphs = "auto" if str(hg).lower() == "auto" else bool(hg)
        self._graphs = None
        self._plan = Plan(config, with_image=self.fusion_enabled, cf=self.cf, image_arch=str(fu.get("image_stream", "resnet18")))
        self._backend = None
        self._build_parameters()
        self.reset_parameters(zero_init_last=bool(fu.get("zero_init_last", False)))
        from .ops import GridSpec
        self._grid = GridSpec(config)

    def _on_moved(self):
        self._backend = None
        self._graphs = None
        self._plan._anc_key = None

    def set_grid(self, voxel_length, voxel_width):
        """Re-plan for another BEV grid IN PLACE.  No parameter depends on the grid (model.py:140-157: convolutions only), so the
        flat parameter / gradient / buffer arenas, every nn.Parameter object (optimizers, hooks and requires_grad flags created
        before the first forward keep pointing at live storage) and the backend with its weight images stay; what depends on the
        grid -- the anchor tensor, the voxel-index affine map, captured graphs -- is rebuilt."""
        L, W = int(voxel_length), int(voxel_width)
        if (L, W) == (self.config["voxel_length"], self.config["voxel_width"]):
            return
        if L % 16 or W % 16:
            raise ValueError("voxel_length and voxel_width must be multiples of 16 (FPN add, model.py:151)")
        cfg = dict(self.config, voxel_length=L, voxel_width=W)
        from .ops import GridSpec
        grid = GridSpec(cfg)                         # validates the new grid before anything is changed
        self.config = cfg
        self._plan.cfg = cfg
        self._plan._anc_key = None
        self._grid = grid
        self._graphs = None

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference's checkpoints with or without DDP's 'module.' prefix (train.py:79)."""
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        return super(ObjectDetection_DCF, self).load_state_dict(sd, strict=strict)

    # ------------------------------------------------------------------ forward
    def _ensure_backend(self, device):
        if self._backend is None or self._backend.dev != device:
            if device.type != "cuda":
                raise H.DcfError("ObjectDetection_DCF runs on the HIP device only: move the module with .cuda() "
                                 "(no CPU fallback exists for the hot path)")
            from .backend_hip import HipBackend
            self._backend = HipBackend(self._plan, self._flat, self._gradflat, self._bufflat, self.dtype,
                                       fp8_min_cin=int(self.config.get("fp8_min_cin", 128)) if self.fp8 else 0,
                                       fp8_min_blocks=int(self.config.get("fp8_min_blocks", 512)))
            if self.config.get("conv_chain") is not None:          # residual stages as chain launches (exclusive use of the GPU only)
                self._backend.chain_enabled = bool(self.config["conv_chain"])
        return self._backend

    KNN_SHARED_MAX_PIXELS = 20000      # sites up to this many pixels (stride 8 and 16 at cfg2) are searched on the finest site's cells
    knn_shared = True
    knn_merged = os.environ.get("DCF_KNN_MERGED", "1") != "0"      # the four sites' cell sorts in one launch per phase

    def fusion_geometry(self, points, uv, n_valid, bufs=None):
        """KNN indices of every fusion site for a batch: points [B,n_max,3], uv [B,n_max,2], n_valid [B] (int).
        bufs: optional persistent buffers dict(idx=[4 x int32 [B,K,h,w]], ws=[4 x workspace]) to write into."""
        from . import ops
        B = points.shape[0]
        dev = points.device
        cnt = n_valid.to(device=dev, dtype=torch.int32).contiguous() if isinstance(n_valid, torch.Tensor) else \
            torch.tensor([int(v) for v in n_valid], dtype=torch.int32, device=dev)
        L, W = self.config["voxel_length"], self.config["voxel_width"]
        idx = []
        batched = points.is_contiguous() and cnt.is_contiguous()
        if batched and self.knn_shared and self.knn_merged:
            # all four sites in one call: their cell sorts share one launch per phase (6 launches instead of 24)
            sites, first = [], None
            for si in range(1, 5):
                s = 2 ** si
                h, w = L // s, W // s
                out = bufs["idx"][si - 1] if bufs is not None else torch.empty((B, self.K, h, w), dtype=torch.int32, device=dev)
                ws = bufs["ws"][si - 1] if bufs is not None else torch.empty((B, ops.knn_ws_stride(points.shape[1], h, w)), dtype=torch.uint8, device=dev)
                fine = first if (first is not None and h * w <= self.KNN_SHARED_MAX_PIXELS) else -1
                if first is None:
                    first = si - 1
                sites.append((h, w, s, fine, ws, out))
            idx = ops.knn_bev_sites(points, cnt, self.K, sites, self._grid.aff, self.r_max)
            return dict(xyz=points.contiguous(), uv=uv.contiguous(), cnt=cnt, idx=idx, aff=self._grid.aff)
        fine = None                     # (h, w, stride, workspace) of the finest site: the coarse sites search ITS cells
        for si in range(1, 5):
            s = 2 ** si
            h, w = L // s, W // s
            if bufs is not None:            # frames land side by side in the batch tensor: no per-frame allocation, no stack copy
                site = bufs["idx"][si - 1]
            else:
                site = torch.empty((B, self.K, h, w), dtype=torch.int32, device=dev)
            ws = None if bufs is None else bufs["ws"][si - 1]
            if batched and fine is not None and h * w <= self.KNN_SHARED_MAX_PIXELS and self.knn_shared:
                # coarse site: own cell sort, but the pixels of dense regions are served from the finest site's cells
                ops.knn_bev_batch_shared(points, cnt, self.K, h, w, s, fine[:3], fine[3], self._grid.aff, self.r_max, ws=ws, out=site)
            elif batched and (B > 1 or self.knn_shared):
                # every phase once for the whole batch (grid.y = frame): half the launches, and the coarse sites' searches fill the chip
                if ws is None:
                    ws = torch.empty((B, ops.knn_ws_stride(points.shape[1], h, w)), dtype=torch.uint8, device=dev)
                ops.knn_bev_batch(points, cnt, self.K, h, w, s, self._grid.aff, self.r_max, ws=ws, out=site)
                if fine is None:
                    fine = (h, w, s, ws)
            else:
                for b in range(B):
                    ops.knn_bev(points[b], cnt[b:b + 1], self.K, h, w, s, self._grid.aff, self.r_max, ws=None if ws is None else ws[0], out=site[b])
            idx.append(site)
        return dict(xyz=points.contiguous(), uv=uv.contiguous(), cnt=cnt, idx=idx, aff=self._grid.aff)

    def fusion_buffers(self, B, n_max, device):
        """Persistent buffers for fusion_geometry / fusion_inverse of batches of B frames with n_max point rows."""
        from . import ops
        L, W = self.config["voxel_length"], self.config["voxel_width"]
        idx, ws = [], []
        for si in range(1, 5):
            s = 2 ** si
            idx.append(torch.empty((B, self.K, L // s, W // s), dtype=torch.int32, device=device))
            ws.append(torch.empty((B, ops.knn_ws_stride(n_max, L // s, W // s)), dtype=torch.uint8, device=device))
        maps = [t[b] for t in idx for b in range(B)]
        ns, ne, nw = ops.fusion_invert_sizes(maps, n_max)
        inv = (torch.empty((ns,), dtype=torch.int32, device=device), torch.empty((2, ne), dtype=torch.int32, device=device),
               torch.empty((nw,), dtype=torch.uint8, device=device))
        return dict(idx=idx, ws=ws, inv=inv)

    def fusion_inverse(self, geom, bufs=None):
        """The fusion backward gathers by POINT: invert every site's KNN map (pairs sorted by point id).  Only the
        backward needs it, so a caller with a side stream records its own event after this (train.geometry_async)."""
        from . import ops
        from .engine import FUSION_INV
        if FUSION_INV == "0":
            return geom
        n_max = geom["xyz"].shape[1]
        maps = [t[b] for t in geom["idx"] for b in range(t.shape[0])]          # map index = site * B + frame
        geom["inv"] = ops.fusion_invert(maps, n_max, out=None if bufs is None else bufs["inv"])
        geom["inv_nmax"] = n_max
        return geom

    def forward(self, x_lidar, x_image, points=None, uv=None, n_valid=None, geom=None):
        """geom: optional result of fusion_geometry() computed ahead of time (e.g. on a side stream, see
        train.Train.geometry_async); otherwise it is derived here from points / uv / n_valid."""
        K = self._ensure_backend(x_lidar.device)
        if not self.fusion_enabled:
            if geom is not None and geom.get("voxel_event") is not None:
                torch.cuda.current_stream().wait_event(geom["voxel_event"])
            geom = None
        elif geom is None and points is not None:
            geom = self.fusion_geometry(points, uv, n_valid)
            if torch.is_grad_enabled():
                self.fusion_inverse(geom)
        bn_train = self.bn_mode == "train" or (self.bn_mode == "module" and self.training)
        if K.set_bn_mode(bn_train):
            self._graphs = None          # captured graphs describe the other BatchNorm mode (other layer table, other kernels)
        if bn_train:
            self._nbtflat += 1
        need = torch.is_grad_enabled() and self._param_list[0].requires_grad
        if self.graphs_wanted(x_lidar.shape[0]) and need and not self._profiling():
            if self._graphs is None:
                self._graphs = _StepGraphs(self)
            if self._graphs.misses > self._graphs.MAX_MISSES and not self._graphs.disabled:
                import warnings
                warnings.warn("hip_graphs: more than %d input signatures captured -- falling back to eager launches" % self._graphs.MAX_MISSES)
                self._graphs.disabled = True
                self._graphs.sets.clear()
            if not self._graphs.disabled:
                return _RunGraphs.apply(self._param_list[0], self, x_lidar, x_image, geom)
        K.prepare()
        return _RunPlan.apply(self._param_list[0], self, x_lidar, x_image, geom, need)

    def graphs_wanted(self, batch):
        if self.use_graphs == "auto":
            # the launch-bound case of a single-rank run: one frame per step.  (Train-mode BatchNorm can be captured too -- set
            # hip_graphs: true -- but its ~600 launches per cfg2 step are paced by the GPU's own per-kernel dispatch, not by the
            # host: 8.06 ms replayed against 7.73 ms eager, profiles/r05n_timeline_trainbn.txt.)
            import torch.distributed as dist
            multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
            return batch == 1 and not multi
        return bool(self.use_graphs)

    graphs_off = False     # set True to force the eager path (per-kernel event timing cannot see inside a graph)

    def _profiling(self):
        return self.graphs_off


class _RunStack(torch.autograd.Function):
    """autograd bridge of a StackPlan (same role as _RunPlan): several outputs, gradient for the input too."""

    @staticmethod
    def forward(ctx, x, token, module, need):
        ctx.module, ctx.saved_graph = module, need
        outs = module._plan.forward(module._backend, x, save=need)
        module._fwd_serial = ctx.serial = getattr(module, "_fwd_serial", 0) + 1
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        m = ctx.module
        if not ctx.saved_graph:
            raise RuntimeError("backward through a forward that ran without saving activations")
        if ctx.serial != m._fwd_serial:
            raise RuntimeError("backward of a stale forward: this module keeps the activations of its LAST forward only "
                               "(call backward before the next forward; gradients are overwritten, not accumulated)")
        gx = m._plan.backward(m._backend, list(gouts))
        m._bind_grads()
        return gx, None, None, None


class _ResidualStack(_FlatParamModule):
    """Common body of the reference's three backbone building blocks (model.py:10-79) on the HIP engine: NCHW fp32 in,
    NCHW fp32 out, the reference's parameter names, forward and backward through the same Block kernels as the full network.
    compute dtype: class attribute `dtype` ("f32" | "bf16" | "f16"), or the `dtype=` keyword; BatchNorm follows
    .train()/.eval() like nn.BatchNorm2d (bn_mode "module"), or is pinned with bn_mode "eval" / "train"."""
    dtype = "f32"
    bn_mode = "module"

    def _setup(self, stages, taps, dtype=None, bn_mode=None):
        if dtype is not None:
            self.dtype = dtype
        if bn_mode is not None:
            self.bn_mode = bn_mode
        self._plan = StackPlan(stages, taps)
        self._backend = None
        self._build_parameters()
        self.reset_parameters()

    def _ensure_backend(self, device):
        if self._backend is None or self._backend.dev != device:
            if device.type != "cuda":
                raise H.DcfError("%s runs on the HIP device only: move the module with .cuda() (no CPU fallback exists "
                                 "for the hot path)" % type(self).__name__)
            from .backend_hip import HipBackend
            self._backend = HipBackend(self._plan, self._flat, self._gradflat, self._bufflat, H.dtype_code(self.dtype))
        return self._backend

    def _run(self, x):
        K = self._ensure_backend(x.device)
        bn_train = self.bn_mode == "train" or (self.bn_mode == "module" and self.training)
        K.set_bn_mode(bn_train)
        if bn_train:
            self._nbtflat += 1
        need = torch.is_grad_enabled() and (self._param_list[0].requires_grad or x.requires_grad)
        K.prepare()
        return _RunStack.apply(x.contiguous(), self._param_list[0], self, need)


class ResidualBlock(_ResidualStack):
    """model.py:10-45: relu(bn2(conv2(relu(bn1(conv1 x)))) + shortcut(x)); stride 2 and a 1x1 shortcut convolution exactly
    when in_channels != out_channels.  Keys: conv1, bn1, conv2, bn2 (, down_conv, down_bn)."""

    def __init__(self, in_channels, out_channels, dtype=None, bn_mode=None):
        super(ResidualBlock, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self._setup([[("", in_channels, out_channels)]], [0], dtype, bn_mode)

    @property
    def should_apply_shortcut(self):
        return self.in_channels != self.out_channels

    def forward(self, x):
        return self._run(x)[0]


class ResidualBlockModule(_ResidualStack):
    """model.py:48-61: num_resblock ResidualBlocks, the first one first_in_channel -> last_out_channel.
    Keys: sequential.resblock_<i>.*"""

    def __init__(self, first_in_channel, last_out_channel, num_resblock, dtype=None, bn_mode=None):
        super(ResidualBlockModule, self).__init__()
        blocks = [("sequential.resblock_%d." % i, first_in_channel if i == 0 else last_out_channel, last_out_channel)
                  for i in range(num_resblock)]
        self._setup([blocks], [0], dtype, bn_mode)

    def forward(self, x):
        return self._run(x)[0]


class ResnetCustomed(_ResidualStack):
    """model.py:64-79: five ResidualBlockModules; returns (x4, x3, x2) = the outputs of layer5, layer4, layer3.
    Keys: layer<k>.sequential.resblock_<i>.*"""

    def __init__(self, out_feature=(32, 64, 128, 192, 256), num_res_block=(1, 2, 4, 6, 6), dtype=None, bn_mode=None):
        super(ResnetCustomed, self).__init__()
        stages = []
        for k in range(5):
            cin = out_feature[0] if k == 0 else out_feature[k - 1]
            stages.append([("layer%d.sequential.resblock_%d." % (k + 1, i), cin if i == 0 else out_feature[k], out_feature[k])
                           for i in range(num_res_block[k])])
        self._setup(stages, [4, 3, 2], dtype, bn_mode)

    def forward(self, x):
        x4, x3, x2 = self._run(x)
        return x4, x3, x2


class OffsettoBbox(nn.Module):
    """model.py:116-137 -- kept for API compatibility; the engine fuses the decode into the
    head kernel (dcf_head_fwd).  Standalone use decodes a [B,14,h,w] offset tensor on the device."""

    def __init__(self, config):
        super(OffsettoBbox, self).__init__()
        self.anchor_bbox_feature = AnchorBoundingBoxFeature(config)

    def forward(self, x):
        from . import ops
        B, C, h, w = x.shape
        head = torch.zeros((B, 32, h, w), dtype=torch.float32, device=x.device)
        head[:, 4:18] = x
        nhwc = ops.nchw_to_nhwc(head.contiguous(), H.F32)
        pred = ops.head_fwd(H.F32, nhwc, self.anchor_bbox_feature().to(x.device))
        return pred[:, 18:32]


class LidarBackboneNetwork(nn.Module):
    """model.py:140-173 surface: (x_cls [B,4,h,w], x_reg [B,14,h,w]) = net(x [B,C,L,W]).
    Implemented as a view of the full engine (the decode channels are simply dropped).

    Like the reference's, the constructor needs no config (`LidarBackboneNetwork()`, model.py:139): the parameters do not
    depend on the BEV grid, only the engine's plan does, so the net is built on the packaged config's grid and re-planned
    IN PLACE (`ObjectDetection_DCF.set_grid`: same arenas, same nn.Parameter objects) for the grid of an input that differs (L and W multiples of 16, model.py:151)."""

    def __init__(self, out_feature=(32, 64, 128, 192, 256), num_res_block=(1, 2, 4, 6, 6), Num_anchor=2, config=None):
        super(LidarBackboneNetwork, self).__init__()
        if Num_anchor != 2:
            raise NotImplementedError("the head kernel is specialised for the reference's 2 anchors")
        if config is None:
            import yaml
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "config_carla.yaml")) as f:
                cfg = yaml.safe_load(f)
            cfg["voxel_channel"] = out_feature[0]
        else:
            cfg = dict(config)
        cfg["lidar_module"] = dict(("out_feature%d" % (i + 1), out_feature[i]) for i in range(5))
        cfg["lidar_module"].update(dict(("num_res_block%d" % (i + 1), num_res_block[i]) for i in range(5)))
        cfg["fusion"] = {"enabled": False}
        self._cfg = cfg
        self.net = ObjectDetection_DCF(cfg)

    def forward(self, x):
        L, W = int(x.shape[2]), int(x.shape[3])
        if (L, W) != (self._cfg["voxel_length"], self._cfg["voxel_width"]):
            # same module, same Parameter objects: an optimizer built before this forward keeps training the live weights
            self.net.set_grid(L, W)
            self._cfg = self.net.config
        pred = self.net(x, None)
        return pred[:, 0:4], pred[:, 4:18]
