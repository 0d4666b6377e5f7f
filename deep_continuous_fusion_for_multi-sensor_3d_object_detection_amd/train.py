"""Drop-in counterpart of the reference's train.py: Train(config).one_step / get_loss_value.

One process per GPU (torchrun); gradients live in the model's flat arena, so data parallelism
is ONE RCCL all-reduce of that arena per step followed by ONE fused Adam launch -- instead of
the reference's single-process DDP wrapper (train.py:24, which current torch rejects).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .loss import LossTotal
from ._hip import torch_dtype as H_torch_dtype
from .model import ObjectDetection_DCF


class FlatAdam(object):
    """Adam(lr, betas=(beta1, 0.999), eps=1e-8) of train.py:28 on the flat parameter arena."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.step_count = 0
        self.m = torch.zeros_like(model.flat_params)
        self.v = torch.zeros_like(model.flat_params)

    def zero_grad(self):
        pass  # the backward pass overwrites the gradient arena

    def step(self, gscale=1.0):
        self.step_count += 1
        ops.adam_step(self.model.flat_params, self.model.flat_grads, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                      self.eps, self.step_count, gscale)

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _through_host(t):
    """gloo with a device tensor (DCF_DIST_BACKEND=gloo: several ranks sharing one GPU in the functional tests): the
    collective runs on a host copy.  RCCL (the product path) takes the device tensor itself."""
    return t.is_cuda and dist.get_backend() == "gloo"


def allreduce_grads(flat_grads):
    """Sum the gradient arena over ranks (RCCL over xGMI on the GPU box, gloo in the CPU tests)."""
    if world() > 1:
        if _through_host(flat_grads):
            h = flat_grads.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            flat_grads.copy_(h)
        else:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return world()


def broadcast_from_rank0(t):
    if world() > 1:
        if _through_host(t):
            h = t.cpu()
            dist.broadcast(h, 0)
            t.copy_(h)
        else:
            dist.broadcast(t, 0)


class Train(nn.Module):
    def __init__(self, config):
        super(Train, self).__init__()
        self.config = config
        self.loss_total = LossTotal(config)
        self.model = ObjectDetection_DCF(config).cuda()
        self.loss_value = None
        self.optimizer = FlatAdam(self.model, config["learning_rate"], (config["beta1"], 0.999))
        self.sync_replicas()
        self._side = None
        self.static_geometry = bool(config.get("static_geometry", True))
        self._geo_sets, self._geo_slot = {}, 0
        # data parallel: all-reduce the LiDAR + fusion gradient bucket under the camera stream's backward.  True (default) =
        # when the world has more than one rank; "force" = also in a one-rank process group (the functional test of the
        # bucketed path on RCCL with the one GPU a test box has, tests/test_gpu_dp.py); False = one all-reduce at the end.
        ov = config.get("overlap_allreduce", os.environ.get("DCF_OVERLAP_ALLREDUCE", "1") != "0")
        self.overlap_force = ov == "force"
        self.overlap_allreduce = bool(ov)
        # factor applied to every rank's gradient INSIDE the RCCL reduction (ncclRedOp PreMulSum) and undone in the Adam step's
        # gradient scale -- e.g. a power of two that lifts small fp16 gradients over the exchange; None = plain sum
        self.allreduce_premul = config.get("allreduce_premul", None)
        # grad_bucket_dtype: "bf16" exchanges the gradient buckets as bf16 (half the bytes over xGMI: 48 MB instead of 96 MB per
        # step at cfg2; every rank rounds its own gradient once, the sum of the rounded values is widened back to fp32 for Adam);
        # "f32" (default) exchanges the fp32 arena itself
        self.grad_bucket_dtype = str(config.get("grad_bucket_dtype", os.environ.get("DCF_GRAD_BUCKET_DTYPE", "f32"))).lower()
        if self.grad_bucket_dtype not in ("f32", "fp32", "bf16"):
            raise ValueError("grad_bucket_dtype must be f32 or bf16 (got %r)" % (self.grad_bucket_dtype,))
        self._g16 = None
        self._pending, self._reduced, self._widen = [], 0, []

    def sync_replicas(self):
        """Identical replicas: rank 0's parameters, buffers, optimiser moments and step count win (called at construction;
        call it again after loading weights or a checkpoint on rank 0 only)."""
        for t in (self.model.flat_params, self.model._bufflat, self.optimizer.m, self.optimizer.v):
            broadcast_from_rank0(t)
        if world() > 1:                          # Adam's bias correction depends on it
            sc = torch.tensor([self.optimizer.step_count], dtype=torch.int64)
            if dist.get_backend() != "gloo":
                sc = sc.to(self.model.flat_params.device)
            dist.broadcast(sc, 0)
            self.optimizer.step_count = int(sc.item())

    def _geo_set(self, B, mp, fast, dims):
        """Persistent device buffers of the per-step geometry, two sets used in turn: a step's geometry is produced on the side
        stream while the previous step's backward may still read its own set.  One zero-fill per step (the projection
        outputs are dense-packed and zero-padded, data_import_carla.py:263-266) instead of a dozen allocations + fills, and
        fixed addresses, so that captured graphs can read the geometry without staging copies."""
        self._geo_slot ^= 1
        key = (B, mp, fast, self._geo_slot)
        st = self._geo_sets.get(key)
        if st is None:
            Cz, L, W = dims
            dev = self.model.flat_params.device
            st = {"slot": self._geo_slot, "free_event": None}
            if fast:
                from ._hip import torch_dtype
                st["x_lidar"] = torch.empty((B, L, W, Cz), dtype=torch_dtype(self.model.dtype), device=dev)
            else:
                st["x_lidar"] = torch.empty((B, Cz, L, W), dtype=torch.float32, device=dev)
            st["proj"] = torch.zeros((B * mp * 5 + 16,), dtype=torch.float32, device=dev)        # [xyz | uv | counts]
            st["xyz"] = st["proj"][:B * mp * 3].view(B, mp, 3)
            st["uv"] = st["proj"][B * mp * 3:B * mp * 5].view(B, mp, 2)
            st["cnt"] = st["proj"][B * mp * 5:B * mp * 5 + B].view(torch.int32)
            st["cnt_host"] = torch.empty(B, dtype=torch.int32).pin_memory()
            if self.model.fusion_enabled:
                st["fusion"] = self.model.fusion_buffers(B, mp, dev)
            self._geo_sets[key] = st
        return st

    def geometry_async(self, frame_geometry, points_list, crts=None, wait_event=None):
        """Per-frame geometry (voxelise, project, KNN of the fusion sites) on a side HIP stream, so that these
        small latency-bound kernels overlap the camera stream's convolutions on the compute stream.
        Returns (x_lidar [B,Cz,L,W], geom) where geom carries the events the engine waits on.
        crts: optional per-frame [4,3] projection matrices (KITTI calibrates every frame).
        wait_event: event the side stream has to wait for before it reads the points (FrameLoader's H2D copies).
        The returned tensors live in one of two persistent buffer sets (config static_geometry, default on): they stay
        valid until the second-next call."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        if wait_event is not None:
            self._side.wait_event(wait_event)
        Cz, L, W = frame_geometry.grid.dims
        Bn, mp = len(points_list), int(frame_geometry.config["max_num_pc"])
        # 16-bit compute types: the voxeliser writes the engine's input image ([B,L,W,Cz] in that type) directly -- no fp32
        # grid, no transpose (the model recognises it by its dtype); f32 keeps the reference's [B,Cz,L,W] grid
        fast = self.model.dtype != 0 and frame_geometry.voxel_mode == 0 and Bn <= 8
        direct = self.static_geometry and all(p.shape[0] <= mp for p in points_list)
        st = None
        if direct:
            # (allocated under the side stream: memory handed out by the caching allocator is only safe to write on the stream
            # it was requested on -- a block the compute stream has just released may still be in use by its queued kernels)
            with torch.cuda.stream(self._side):
                st = self._geo_set(Bn, mp, fast, (Cz, L, W))
        if st is not None:
            if st["free_event"] is not None:
                self._side.wait_event(st["free_event"])        # the step that last read this set has finished with it
                st["free_event"] = None
            elif st.get("used"):
                self._side.wait_stream(main)                   # handed out before, never consumed by one_step: be conservative
            st["used"] = True
        with torch.cuda.stream(self._side):
            pcs, uvs, cnts = [], [], []
            if st is not None:
                x_lidar = st["x_lidar"]
                st["proj"].zero_()
                xyz_all, uv_all, cnt_all = st["xyz"], st["uv"], st["cnt"]
            else:
                x_lidar = torch.empty((Bn, L, W, Cz), dtype=H_torch_dtype(self.model.dtype), device="cuda") if fast else \
                    torch.empty((Bn, Cz, L, W), dtype=torch.float32, device="cuda")
                xyz_all = torch.zeros((Bn, mp, 3), dtype=torch.float32, device="cuda")
                uv_all = torch.zeros((Bn, mp, 2), dtype=torch.float32, device="cuda")
                cnt_all = torch.zeros((Bn,), dtype=torch.int32, device="cuda")
            # projection first: its valid-point counts go to the host (pinned, asynchronous) while the voxeliser and the KNN
            # still run; the engine sizes the per-point fusion tensors by them instead of max_num_pc (Plan._fusion_rows)
            # the frames' projections land side by side in batch tensors (no per-frame allocations, no stack copies)
            inplace = True
            if getattr(frame_geometry, "project_batch", None) is not None and frame_geometry.project_batch(points_list, crts, xyz_all, uv_all, cnt_all):
                pass                                    # every frame in one launch per phase (count / scan / scatter)
            else:
                for b, pts in enumerate(points_list):
                    pc, uv, cnt = frame_geometry.project(pts, crt=None if crts is None else crts[b],
                                                         out=(xyz_all[b], uv_all[b], cnt_all[b:b + 1]))
                    inplace = inplace and pc.data_ptr() == xyz_all[b].data_ptr()
                    pcs.append(pc); uvs.append(uv); cnts.append(cnt)
            cnt_dev = cnt_all if inplace else torch.cat(cnts, 0)
            cnt_host = st["cnt_host"] if st is not None else torch.empty(Bn, dtype=torch.int32).pin_memory()
            cnt_host.copy_(cnt_dev, non_blocking=True)
            ev_cnt = torch.cuda.Event()
            ev_cnt.record()
            frame_geometry.voxelize_batch(points_list, x_lidar, self.model.dtype if fast else None)   # written in place, frames side by side
            ev_vox = torch.cuda.Event()
            ev_vox.record()
            geom = None
            if self.model.fusion_enabled:
                fb = st.get("fusion") if st is not None else None
                if inplace:
                    geom = self.model.fusion_geometry(xyz_all, uv_all, cnt_all, bufs=fb)
                else:                                   # a frame with more than max_num_pc points: the copying path
                    geom = self.model.fusion_geometry(torch.stack(pcs, 0), torch.stack(uvs, 0), cnt_dev)
                ev = torch.cuda.Event()
                ev.record()
                geom["event"] = ev
                geom["cnt_host"], geom["cnt_event"] = cnt_host, ev_cnt
                self.model.fusion_inverse(geom, bufs=fb if inplace else None)        # needed by the backward only: its own event
                ev_inv = torch.cuda.Event()
                ev_inv.record()
                geom["inv_event"] = ev_inv
            else:
                geom = {}
            geom["voxel_event"] = ev_vox
        if st is not None:
            geom["_set"] = st                # one_step records the set's free event when the step has consumed it
            geom["static"] = inplace
        else:
            # tensors born on the side stream are consumed on the compute stream
            x_lidar.record_stream(main)
            born = [geom.get("xyz"), geom.get("uv"), geom.get("cnt")] + list(geom.get("idx") or [])
            born += list(geom.get("inv") or [])
            for t in born:
                if t is not None:
                    t.record_stream(main)
        return x_lidar, geom

    def _predict(self, lidar_voxel, camera_image, extra):
        pred = self.model(lidar_voxel, camera_image, **extra)
        return torch.split(pred, [4, 14, 14], dim=1)

    def _bucket_ready(self, ranges):
        """Backend hook (world size > 1): arena ranges whose gradients are final.  Their all-reduce starts now -- RCCL runs it
        on its own stream behind the finalisation launch, under whatever the backward still has to do -- and is waited for
        before the optimiser step."""
        g = self.model.flat_grads
        for a, b in ranges:
            if b <= a:
                continue
            seg = g[a:b]
            if self.grad_bucket_dtype == "bf16":
                # the bucket goes out in bf16: rounded here (behind the finalisation launch, same stream), summed by the collective,
                # widened back into the fp32 arena once the collective is done (one_step, before the optimiser step)
                from . import ops
                from . import _hip as H
                if self._g16 is None or self._g16.numel() != g.numel():
                    self._g16 = torch.empty(g.numel(), dtype=torch.bfloat16, device=g.device)
                seg16 = self._g16[a:b]
                H.call("dcf_cast", H.F32, seg, H.BF16, seg16, b - a, H.stream_ptr())
                self._widen.append((a, b))
                if _through_host(seg16):
                    h = seg16.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    seg16.copy_(h)
                else:
                    self._pending.append(dist.all_reduce(seg16, op=dist.ReduceOp.SUM, async_op=True))
            elif _through_host(seg):
                h = seg.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                seg.copy_(h)
            else:
                op = dist.ReduceOp.SUM if self.allreduce_premul is None else dist._make_nccl_premul_sum(float(self.allreduce_premul))
                self._pending.append(dist.all_reduce(seg, op=op, async_op=True))
            self._reduced += b - a

    def one_step(self, lidar_voxel, camera_image, object_data, num_ref_box, **extra):
        pred_cls, pred_reg, _ = self._predict(lidar_voxel, camera_image, extra)
        self.loss_value = self.loss_total(object_data, num_ref_box, pred_cls, pred_reg)
        self.optimizer.zero_grad()
        n = world()
        grouped = dist.is_available() and dist.is_initialized()
        overlap = ((n > 1 or (self.overlap_force and grouped)) and self.overlap_allreduce and not self.model.graphs_wanted(lidar_voxel.shape[0])
                   and self.model._backend is not None)
        premul = 1.0
        self._pending, self._reduced, self._widen = [], 0, []
        if overlap:
            self.model._backend.bucket_hook = self._bucket_ready
        try:
            self.loss_value.backward()
        finally:
            if overlap:
                self.model._backend.bucket_hook = None
        if overlap and self._reduced == self.model.flat_grads.numel():
            for w in self._pending:
                w.wait()
            if self._widen:                         # bf16 buckets: the summed values back into the fp32 arena
                from . import _hip as H
                for a, b in self._widen:
                    H.call("dcf_cast", H.BF16, self._g16[a:b], H.F32, self.model.flat_grads[a:b], b - a, H.stream_ptr())
            elif self.allreduce_premul is not None and not _through_host(self.model.flat_grads):
                premul = float(self.allreduce_premul)
        else:                                      # single rank, captured graphs, or a backward that skipped the buckets
            for w in self._pending:
                w.wait()
            if self._reduced:
                raise RuntimeError("gradient buckets covered %d of %d elements" % (self._reduced, self.model.flat_grads.numel()))
            allreduce_grads(self.model.flat_grads)
        self.optimizer.step(1.0 / (n * premul))
        st = (extra.get("geom") or {}).get("_set")
        if st is not None:                         # the geometry buffers of this step may be refilled from here on
            st["free_event"] = torch.cuda.Event()
            st["free_event"].record()

    def one_step_raw(self, frame_geometry, batch):
        """One train step from a FrameLoader batch (raw points + image in HBM): geometry on the side stream, then one_step."""
        batch.wait()
        x_lidar, geom = self.geometry_async(frame_geometry, batch["points"], crts=batch.get("crt"), wait_event=batch.event)
        self.one_step(x_lidar, batch["image"], batch["bboxes"], batch["num_bboxes"], geom=geom)

    # ------------------------------------------------------------------ checkpoint / resume
    def save_checkpoint(self, path, epoch=0):
        """Model state_dict (the reference's key names, train.py:79) plus what the reference never saved:
        optimiser moments / step count and the epoch, so that training can really resume."""
        sd = {k: v.detach().clone().contiguous().cpu() for k, v in self.model.state_dict().items()}
        opt = {"step": self.optimizer.step_count, "m": self.optimizer.m.cpu(), "v": self.optimizer.v.cpu()}
        torch.save({"model": sd, "optimizer": opt, "epoch": int(epoch), "loss_calls": int(getattr(self.loss_total, "calls", 0))}, path)

    def load_checkpoint(self, path):
        ck = torch.load(path, map_location="cpu")
        if "model" not in ck:                      # a bare state_dict written by the reference's train.py
            self.model.load_state_dict(ck)
            return 0
        self.model.load_state_dict(ck["model"])
        self.optimizer.load_state_dict({k: (v.to(self.model.flat_params.device) if torch.is_tensor(v) else v)
                                        for k, v in ck["optimizer"].items()})
        if hasattr(self.loss_total, "calls"):       # device sampling: a resumed run continues the draw sequence instead of replaying it
            self.loss_total.calls = int(ck.get("loss_calls", self.optimizer.step_count))
        return int(ck.get("epoch", 0))

    def get_loss_value(self, lidar_voxel, camera_image, object_data, num_ref_box, **extra):
        with torch.no_grad():
            pred_cls, pred_reg, _ = self._predict(lidar_voxel, camera_image, extra)
            self.loss_value = self.loss_total(object_data, num_ref_box, pred_cls, pred_reg)
        return self.loss_value.item(), pred_cls, pred_reg


_HOST_PG = None        # gloo side group of init_distributed(): host-side fences that never touch RCCL


def _eval_barrier(timeout_s=None):
    """All ranks meet before and after rank 0's evaluation (train.main): no rank enters a gradient all-reduce while rank 0 is
    still evaluating.  The device is drained first; the fence itself is a HOST barrier on the gloo side group
    (dist.monitored_barrier: TCP store round trips, its own timeout, names the rank that did not arrive) -- with the product's
    `nccl` backend a plain dist.barrier() is itself an RCCL all-reduce, i.e. the waiting ranks would sit in a pending collective
    under the RCCL watchdog for as long as rank 0 evaluates.  Timeout: DCF_EVAL_BARRIER_TIMEOUT_S (default
    3600 s, an evaluation pass over the whole test set)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if _HOST_PG is not None:
        import datetime
        t = float(timeout_s if timeout_s is not None else os.environ.get("DCF_EVAL_BARRIER_TIMEOUT_S", "3600"))
        dist.monitored_barrier(group=_HOST_PG, timeout=datetime.timedelta(seconds=t))
    else:
        dist.barrier()          # the default group is gloo already (functional runs) or no side group could be made


def init_distributed():
    """env:// rendezvous, one rank per GPU (RANK/LOCAL_RANK/WORLD_SIZE from torchrun)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    # (DCF_FORCE_DIST=1: a process group even for one rank -- the RCCL path on a one-GPU test box)
    if (ws > 1 or os.environ.get("DCF_FORCE_DIST") == "1") and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm (xGMI). DCF_DIST_BACKEND=gloo lets two ranks share one GPU for functional tests.
        backend = os.environ.get("DCF_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    global _HOST_PG
    if dist.is_available() and dist.is_initialized() and _HOST_PG is None and dist.get_world_size() > 1:
        # host-side fences (evaluation barriers) go through a gloo group of their own: created once, by every rank, here
        _HOST_PG = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD
    return ws


def make_dataset(config, mode="train"):
    """train.py:58-65: CarlaDataset or KittiDataset by config["dataset_name"], in raw mode for the FrameLoader;
    synthetic frames when the data directory does not exist."""
    from .data_import_carla import CarlaDataset, SyntheticDataset
    from .data_import_kitti import KittiDataset
    if config.get("dataset_name") == "kitti":
        ds = KittiDataset(config, mode=mode, raw=True)
        if len(ds):
            return ds
    elif os.path.isdir(config["train_data_dir" if mode == "train" else "test_data_dir"]):
        ds = CarlaDataset(config, mode=mode, raw=True)
        if len(ds):
            return ds
    print("no %s frames under the configured data directory: synthetic frames" % mode)
    return SyntheticDataset(config, length=64, raw=True)


def evaluate(training, tester, dataset, loader, max_batches=None):
    """The reference's evaluation pass (train.py:87-100 every 500 batches, :105-122 at the end of an epoch): loss, the
    score-threshold / NMS / precision-recall counters of test.Test over the test loader.  Returns (mean loss, cumulative
    loss, number of positives, number of labelled boxes, TP counters per IoU threshold)."""
    tester.initialize_ap()
    cum = 0.0
    n = 0
    for n, batch in enumerate(loader, 1):
        batch.wait()
        x_lidar, geom = training.geometry_async(dataset.geometry, batch["points"], crts=batch.get("crt"), wait_event=batch.event)
        value, _ = tester.get_eval_value_onestep(x_lidar, batch["image"], batch["bboxes"], batch["num_bboxes"], geom=geom)
        cum += value
        if max_batches is not None and n >= max_batches:
            break
    tester.display_average_precision()
    return cum / max(n, 1), cum, tester.get_num_P(), tester.get_num_T(), tester.get_num_TP_set()


def main():
    import yaml
    from .frame_loader import FrameLoader
    from .test import Test
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "config", "config_carla.yaml")) as f:
        config = yaml.safe_load(f)
    init_distributed()
    rank0 = not dist.is_initialized() or dist.get_rank() == 0
    dataset = make_dataset(config)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=True) if world() > 1 else None
    loader = FrameLoader(dataset, config["batch_size"], sampler=sampler, shuffle=True, num_workers=int(config.get("num_workers", 4)),
                         drop_last=True)
    training = Train(config)
    # the reference builds Test(training.model) before the loop (train.py:76), which is what puts the trained module in eval
    # mode: bn_mode "eval" is the default here, and with bn_mode "module" this constructor has the same effect
    tester = Test(training.model, config)
    test_dataset = make_dataset(config, "test") if rank0 else None
    test_loader = FrameLoader(test_dataset, config["batch_size"], shuffle=False, num_workers=int(config.get("num_workers", 4)),
                              drop_last=True) if rank0 else None
    os.makedirs("./saved_model", exist_ok=True)
    for epoch in range(config["num_epoch"]):
        if sampler is not None:
            sampler.set_epoch(epoch)
        if rank0:
            torch.save(training.model.state_dict(), "./saved_model/" + config["saved_model_name"])
        for batch_ndx, batch in enumerate(loader):
            training.one_step_raw(dataset.geometry, batch)
            if batch_ndx % 100 == 0:
                print("training at ", batch_ndx, "is processed, loss %.4f" % training.loss_value.item())
            if batch_ndx % 500 == 0 and batch_ndx != 0:                    # train.py:87-100
                # Rank 0 evaluates, the others WAIT HERE: without the barriers they would walk into the next step's gradient
                # all-reduce and sit in it for as long as the evaluation takes (minutes: an RCCL watchdog hazard).  A barrier on
                # the host (monitored_barrier on the gloo side group of init_distributed) has no such timeout coupling to a pending collective.
                _eval_barrier()
                if rank0:
                    mean, cum, npos, nt, tp = evaluate(training, tester, test_dataset, test_loader, max_batches=int(config.get("eval_batches", 7)))           # `batch_ndx_ > 5`
                    print("batch %d: validation loss %.4f, positives %d, labelled %d, TP@0.5 %d" % (batch_ndx, mean, npos, nt, tp[0.5]))
                _eval_barrier()
        _eval_barrier()
        if rank0:                                                          # train.py:105-122
            mean, cum, npos, nt, tp = evaluate(training, tester, test_dataset, test_loader, max_batches=int(config.get("eval_batches_epoch", 12)))   # `batch_ndx > 10`
            print("epoch %d: validation loss %.4f (cumulative %.2f), positives %d, labelled %d, TP %s" % (epoch, mean, cum, npos, nt, dict(tp)))
        _eval_barrier()


if __name__ == "__main__":
    main()
