"""FrameLoader: raw frames from worker processes to HBM through pinned staging.

The reference's DataLoader (train.py:68-73, :80-85) voxelises on the CPU inside the dataset and ships a 12.6-72 MB fp32
grid per frame to the GPU.  Here the dataset runs in raw mode -- worker processes only read and decode -- and what
crosses PCIe is the point list (12 B/point) and the uint8 image; the voxeliser / projector / KNN run on the GPU
(Train.geometry_async).

Staging runs on a BACKGROUND THREAD (round 5): it pulls the next host batch, packs it into one of
six pinned staging sets with a GIL-releasing memcpy, enqueues the H2D copies on the copy stream and hands the batch over
through a four-deep queue.  The thread that enqueues the train step only takes a finished Batch out of the queue and, when it
comes back for the next one, records ONE event on its stream ("everything that read this set's device buffers has been
enqueued") that the copy stream waits for before it overwrites the set.  Until round 5 the packing, the copy enqueue and an
event synchronise ran on the enqueue thread itself: ~1 ms of a 5.3 ms step on a host that is not ahead of the GPU.

    loader = FrameLoader(dataset, batch_size, sampler=..., num_workers=4)
    for batch in loader:        # batch["points"]: list of [n_i,3] f32 device tensors, batch["image"]: [B,3,H,W] u8
        batch.wait()            # compute stream waits on the copy stream's event
        x_lidar, geom = trainer.geometry_async(dataset.geometry, batch["points"], crts=batch["crt"], wait_event=batch.event)

A batch's device tensors are views of its staging set: they stay valid until the loader is asked for the NEXT batch (the
usual `for` loop), not longer.  With device=None (no GPU: unit tests, host-side tooling) the same batches come back as host
tensors; `threaded=False` stages on the calling thread (one batch ahead, two sets: the pre-round-5 behaviour, kept for A/B runs).
"""
import ctypes
import queue
import threading

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
    they stay valid until the consumer comes back for the NEXT batch (threaded staging: the set then returns to the worker behind
    an event on the consumer's stream; inline staging: until the batch after the next one)."""
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
        self.consumed = None                   # event on the consumer's stream: every reader of the device twins has been enqueued
        self.p_ptr, self.i_ptr = self.points.data_ptr(), self.image.data_ptr()

    def pack(self, host, cap):
        """Host batch -> pinned buffers.  ctypes.memmove releases the GIL for the duration of the copy, so the thread that
        enqueues the train step keeps running while megabytes move."""
        pts = []
        for b, p in enumerate(host["points"]):
            p = p.contiguous()
            n = int(p.shape[0])
            if p.dtype != torch.float32 or p.dim() != 2 or p.shape[1] != 3:
                raise ValueError("raw lidar_points must be float32 [n,3] (got %s %s)" % (p.dtype, tuple(p.shape)))
            ctypes.memmove(self.p_ptr + b * cap * 12, p.data_ptr(), n * 12)
            pts.append((b * cap, n))
        isz = self.image[0].numel()
        for b, im in enumerate(host["image"]):
            im = im.contiguous()
            if im.dtype != torch.uint8 or im.numel() != isz:
                raise ValueError("raw image must be uint8 %s (got %s %s)" % (tuple(self.image.shape[1:]), im.dtype, tuple(im.shape)))
            ctypes.memmove(self.i_ptr + b * isz, im.data_ptr(), isz)
        return pts


class FrameLoader(object):
    def __init__(self, dataset, batch_size, sampler=None, shuffle=False, num_workers=0, device="cuda", max_points=None,
                 prefetch_factor=2, drop_last=False, threaded=True):
        if not getattr(dataset, "raw", False):
            raise ValueError("FrameLoader needs a dataset in raw mode (raw=True): it moves points, not voxel grids")
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.device = torch.device(device) if device is not None else None
        if self.device is not None and self.device.type == "cuda" and not torch.cuda.is_available():
            raise RuntimeError("FrameLoader(device='cuda') needs a GPU; pass device=None for host-side batches")
        if self.device is not None and self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())      # (the staging thread selects it by index)
        kw = dict(batch_size=self.batch_size, sampler=sampler, shuffle=(shuffle and sampler is None), num_workers=num_workers,
                  collate_fn=collate_raw, drop_last=drop_last)
        if num_workers > 0:
            kw.update(prefetch_factor=prefetch_factor, persistent_workers=True)
        self.loader = DataLoader(dataset, **kw)
        self.max_points = max_points
        self.threaded = bool(threaded)
        self._sets, self._copy, self._cap = None, None, 0

    def __len__(self):
        return len(self.loader)

    def _ensure_sets(self, host, nsets):
        need = max(int(p.shape[0]) for p in host["points"])
        ishape = tuple(host["image"][0].shape)
        if self._sets is None or len(self._sets) != nsets or self._cap < need or tuple(self._sets[0].image.shape[1:]) != ishape:
            if self._sets is not None:
                torch.cuda.synchronize(self.device)      # (re-sizing mid-run: nothing may still read the old sets)
            cap = max(need, int(self.max_points or 0))
            cap = (cap + 4095) // 4096 * 4096
            self._sets = [_Staging(cap, self.batch_size, ishape, self.device) for _ in range(nsets)]
            self._cap = cap
            fresh = True
        else:
            fresh = False
        if self._copy is None:
            self._copy = torch.cuda.Stream(self.device)
        if fresh:
            # Prime every set: its first pinned -> device copy pays one-time costs inside the runtime (the first asynchronous copy
            # on the copy stream took 45 ms in tools/fromhost_hiccup.py, and some boxes showed a second such pause a batch or
            # two later -- a 60-90 ms step in the middle of a run).  They are paid here, before the first batch is handed out.
            with torch.cuda.stream(self._copy):
                for st in self._sets:
                    st.dev_points.copy_(st.points, non_blocking=True)
                    st.dev_image.copy_(st.image, non_blocking=True)
            self._copy.synchronize()

    def _stage(self, host, slot, consumer_stream=None):
        """Pack one host batch into staging set `slot` and enqueue its H2D copies on the copy stream.
        consumer_stream: (unthreaded form) the stream whose enqueued work must be through before the device twins are
        overwritten; the threaded form waits for the set's `consumed` event instead."""
        st = self._sets[slot]
        if st.done is not None:
            st.done.synchronize()               # the pinned buffers are reused: their previous copies must be out
        B = len(host["points"])
        pts = st.pack(host, self._cap)
        out = Batch(bboxes=host["bboxes"], num_bboxes=host["num_bboxes"], crt=host["crt"])
        if consumer_stream is not None:
            self._copy.wait_stream(consumer_stream)
        elif st.consumed is not None:
            self._copy.wait_event(st.consumed)
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
        out.slot = slot
        return out

    def _iter_inline(self):
        """Staging on the calling thread, one batch of look-ahead over two sets (the pre-round-5 form)."""
        it = iter(self.loader)
        slot = 0
        try:
            first = next(it)
        except StopIteration:
            return
        self._ensure_sets(first, 2)
        nxt = self._stage(first, slot, torch.cuda.current_stream(self.device))
        for host in it:
            cur, slot = nxt, slot ^ 1
            self._ensure_sets(host, 2)
            # the device twins were last read by the step of the batch two back: everything enqueued so far (that step
            # included; the step of the previous batch is not enqueued yet) must be through before they are overwritten
            nxt = self._stage(host, slot, torch.cuda.current_stream(self.device))
            yield cur
        yield nxt

    # staging sets: one being read by the running step, DEPTH staged and waiting, one being packed.  Four batches of look-ahead
    # (~20 ms of steps at cfg2) ride out a staging thread that the host deschedules for a while (a shared box showed single 45 ms
    # steps with two: profiles/r05c_*); the sets are a few MB each.
    DEPTH = 4
    NSETS = DEPTH + 2

    def _iter_threaded(self):
        free = queue.Queue()            # staging-set indices the worker may fill
        ready = queue.Queue(maxsize=self.DEPTH)  # staged batches, then a sentinel: None = end, an exception = re-raise
        stop = threading.Event()
        dev = self.device

        def put(item):
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def take_free():
            while not stop.is_set():
                try:
                    return free.get(timeout=0.05)
                except queue.Empty:
                    pass
            return None

        def work():
            try:
                torch.cuda.set_device(dev)
                have = False
                for host in self.loader:
                    if stop.is_set():
                        return
                    if not have:
                        self._ensure_sets(host, self.NSETS)
                        for i in range(self.NSETS):
                            free.put(i)
                        have = True
                    elif max(int(p.shape[0]) for p in host["points"]) > self._cap or tuple(host["image"][0].shape) != tuple(self._sets[0].image.shape[1:]):
                        # a larger frame than the sets hold: take every set back (the consumer keeps draining `ready` and
                        # releasing), then re-size with nothing in flight
                        for _ in range(self.NSETS):
                            if take_free() is None:
                                return
                        self._ensure_sets(host, self.NSETS)
                        for i in range(self.NSETS):
                            free.put(i)
                    slot = take_free()
                    if slot is None or not put(self._stage(host, slot)):
                        return
                put(None)
            except BaseException as e:          # handed to the consumer, which re-raises it
                put(e)

        th = threading.Thread(target=work, name="FrameLoader-staging", daemon=True)
        th.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
                # the caller is back for the next batch: every launch that reads this batch's device buffers has been enqueued
                # (the geometry side stream is joined by the compute stream before the step ends) -- one event on the caller's
                # stream, then the set goes back to the worker, whose copy stream waits for that event before overwriting it
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                self._sets[item.slot].consumed = ev
                free.put(item.slot)
        finally:
            stop.set()
            th.join(timeout=5.0)
            if th.is_alive():
                # (a reader that takes longer than this: the worker may still touch the sets -- a new iteration has to wait for it)
                self._straggler = th
            # An iteration that ends early (train.evaluate() breaks at max_batches) leaves batches staged but never handed over, and
            # the last one handed over without its `consumed` event: the next iteration's copy stream must not overwrite a device twin
            # the previous consumer's kernels may still read.  One event on the consumer's stream, now, covers everything it enqueued.
            if self._sets:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                for st in self._sets:
                    st.consumed = ev

    def __iter__(self):
        if self.device is not None and getattr(self, "_active", False):
            raise RuntimeError("FrameLoader: one iteration at a time (the staging sets belong to the running one; close it or finish it first)")
        old = getattr(self, "_straggler", None)
        if old is not None:
            old.join(timeout=60.0)
            if old.is_alive():
                raise RuntimeError("FrameLoader: the previous iteration's staging thread is still inside the dataset reader; its staging sets cannot be reused yet")
            self._straggler = None
        if self.device is None:
            for host in self.loader:
                host["image"] = torch.stack(host["image"], 0)
                yield Batch(host)
            return
        self._active = True
        try:
            yield from (self._iter_threaded() if self.threaded else self._iter_inline())
        finally:
            self._active = False
