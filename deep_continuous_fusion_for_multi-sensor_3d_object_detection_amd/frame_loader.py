"""FrameLoader: raw frames from worker processes to HBM through pinned staging.

The reference's DataLoader (train.py:68-73) voxelises on the CPU inside the dataset and ships a 12.6-72 MB fp32
grid per frame to the GPU.  Here the dataset runs in raw mode -- worker processes only read and decode -- and what
crosses PCIe is the point list (12 B/point) and the uint8 image; the voxeliser / projector / KNN run on the GPU
(Train.geometry_async).  Two sets of pinned staging buffers alternate, the H2D copies go on a copy stream, and the
batch handed to the caller carries the event the compute stream has to wait for, so the copies of batch i+1 overlap
the step of batch i.

    loader = FrameLoader(dataset, batch_size, sampler=..., num_workers=4)
    for batch in loader:        # batch["points"]: list of [n_i,3] f32 device tensors, batch["image"]: [B,3,H,W] u8
        batch.wait()            # compute stream waits on the copy stream's event
        x_lidar, geom = trainer.geometry_async(dataset.geometry, batch["points"], crts=batch["crt"], wait_event=batch.event)

With device=None (no GPU: unit tests, host-side tooling) the same batches come back as host tensors.
"""
import torch
from torch.utils.data import DataLoader


def collate_raw(samples):
    """Raw samples -> one batch dict; point lists and images stay lists (they are packed straight into the pinned
    staging set), the small label tensors are stacked."""
    out = {"points": [s["lidar_points"] for s in samples],
           "image": [s["image"] for s in samples],
           "bboxes": torch.stack([s["bboxes"] for s in samples], 0),
           "num_bboxes": torch.tensor([int(s["num_bboxes"]) for s in samples]),
           "crt": [s["crt"] for s in samples] if samples and samples[0].get("crt") is not None else None}
    return out


class Batch(dict):
    """dict of tensors + the copy-stream event guarding them.  The device tensors are views of the loader's staging set:
    they stay valid until the loader has handed out the batch after the next one."""
    event = None

    def wait(self, stream=None):
        """Make `stream` (default: the current one) wait for the H2D copies of this batch."""
        if self.event is not None:
            (stream or torch.cuda.current_stream()).wait_event(self.event)
        return self


class _Staging(object):
    """One staging set: pinned host buffers (points of a whole batch back to back, the image batch) and their device
    twins -- allocated once, so a step allocates nothing and frees nothing on the copy stream."""

    def __init__(self, max_points, batch, image_shape, device):
        self.points = torch.empty((batch * max_points, 3), dtype=torch.float32).pin_memory()
        self.image = torch.empty((batch,) + tuple(image_shape), dtype=torch.uint8).pin_memory()
        # numpy views for the packing: a plain single-threaded memcpy.  (torch's CPU copy_ of a megabyte-sized tensor opens
        # an OpenMP region whose workers then spin on every core and slow the thread that enqueues the step: measured
        # 35 ms instead of 7 ms per step.)
        self.points_np, self.image_np = self.points.numpy(), self.image.numpy()
        self.dev_points = torch.empty((batch * max_points, 3), dtype=torch.float32, device=device)
        self.dev_image = torch.empty((batch,) + tuple(image_shape), dtype=torch.uint8, device=device)
        self.done = None                       # event: the H2D copies out of this set have finished


class FrameLoader(object):
    def __init__(self, dataset, batch_size, sampler=None, shuffle=False, num_workers=0, device="cuda", max_points=None,
                 prefetch_factor=2, drop_last=False):
        if not getattr(dataset, "raw", False):
            raise ValueError("FrameLoader needs a dataset in raw mode (raw=True): it moves points, not voxel grids")
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.type == "cuda" and not torch.cuda.is_available():
            raise RuntimeError("FrameLoader(device='cuda') needs a GPU; pass device=None for host-side batches")
        kw = dict(batch_size=self.batch_size, sampler=sampler, shuffle=(shuffle and sampler is None), num_workers=num_workers,
                  collate_fn=collate_raw, drop_last=drop_last)
        if num_workers > 0:
            kw.update(prefetch_factor=prefetch_factor, persistent_workers=True)
        self.loader = DataLoader(dataset, **kw)
        self.max_points = max_points
        self._sets, self._copy, self._cap = None, None, 0

    def __len__(self):
        return len(self.loader)

    def _stage(self, host, slot):
        """Pack one host batch into staging set `slot` and enqueue its H2D copies on the copy stream."""
        B = len(host["points"])
        need = max(int(p.shape[0]) for p in host["points"])
        ishape = tuple(host["image"][0].shape)
        if self._sets is None or self._cap < need or tuple(self._sets[0].image.shape[1:]) != ishape:
            cap = max(need, int(self.max_points or 0))
            cap = (cap + 4095) // 4096 * 4096
            self._sets = [_Staging(cap, self.batch_size, ishape, self.device) for _ in range(2)]
            self._cap = cap
        st = self._sets[slot]
        if st.done is not None:
            st.done.synchronize()               # the set is reused every other batch: its previous copies must be out
        if self._copy is None:
            self._copy = torch.cuda.Stream(self.device)
        pts = []
        for b, p in enumerate(host["points"]):
            n = int(p.shape[0])
            st.points_np[b * self._cap:b * self._cap + n] = p.numpy()
            pts.append((b * self._cap, n))
        for b, im in enumerate(host["image"]):
            st.image_np[b] = im.numpy()
        out = Batch(bboxes=host["bboxes"], num_bboxes=host["num_bboxes"], crt=host["crt"])
        # the device twins were last read by the step of the batch two back: everything enqueued so far (that step
        # included; the step of the previous batch is not enqueued yet) must be through before they are overwritten
        self._copy.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._copy):
            out["points"] = []
            for o, n in pts:
                st.dev_points[o:o + n].copy_(st.points[o:o + n], non_blocking=True)
                out["points"].append(st.dev_points[o:o + n])
            st.dev_image[:B].copy_(st.image[:B], non_blocking=True)
            out["image"] = st.dev_image[:B]
            ev = torch.cuda.Event()
            ev.record()
        st.done = ev
        out.event = ev
        return out

    def __iter__(self):
        if self.device is None:
            for host in self.loader:
                host["image"] = torch.stack(host["image"], 0)
                yield Batch(host)
            return
        # one batch of look-ahead: batch i+1 is staged and in flight while the caller works on batch i
        it = iter(self.loader)
        slot = 0
        try:
            nxt = self._stage(next(it), slot)
        except StopIteration:
            return
        for host in it:
            cur, slot = nxt, slot ^ 1
            nxt = self._stage(host, slot)
            yield cur
        yield nxt
