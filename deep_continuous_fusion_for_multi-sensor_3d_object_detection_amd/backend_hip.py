"""HIP backend of the execution plan: arenas, the dcf_conv_param table and the kernel calls.

Owns (per model, per device):
  warena  -- compute-dtype images of every convolution weight, rewritten once per step by
             dcf_weight_prep: [Cout_pad][taps][Cin] for forward/wgrad, [Cin][taps][Cout_pad] for dgrad
  ssarena -- fp32 [scale | shift] per convolution (folded eval BatchNorm)
  gsum    -- fp32 per-channel sums of the masked output gradient (= dL/dbeta)
  slabs   -- fp32 split-K partial weight gradients, reduced in fixed order by dcf_wgrad_finalize
The flat parameter / gradient / buffer arenas belong to the model (model.py).
"""
import ctypes
import os

import numpy as np
import torch

from . import _hip as H
from . import ops

BN_EPS = 1e-5  # nn.BatchNorm2d(eps=1e-05), model.py:20


class HipBackend(object):
    def __init__(self, plan, params, grads, buffers, dtype, fp8_min_cin=0, fp8_min_blocks=512):
        if not params.is_cuda:
            raise H.DcfError("the HIP hot path needs CUDA/HIP tensors (got %s); there is no CPU fallback" % params.device)
        H.lib()  # fail loudly right here when the extension is missing
        self.plan, self.params, self.grads, self.buffers = plan, params, grads, buffers
        self.dtype = H.dtype_code(dtype)
        self.tdtype = H.torch_dtype(self.dtype)
        self.es = 4 if self.dtype == H.F32 else 2
        self.dev = params.device
        self._fus_dws = {}                 # fusion backward: boundary-row workspaces per (pairs, channels, frames)
        self._wtab = {}                    # grouped weight gradients: ctypes tables per layer sequence
        self._chain_ok, self._chain_ws, self._chain_tab = {}, {}, {}     # chain launches: support per shape, arrival counters per shape, ctypes tables per length
        self._fn_fwd, self._fn_dgrad = H.fn("dcf_conv2d_fwd"), H.fn("dcf_conv2d_dgrad")
        self._fn_bnf, self._fn_bnb = H.fn("dcf_bn_train_fwd"), H.fn("dcf_bn_train_bwd")
        self._bbase = buffers.data_ptr()
        self._pbase, self._gbase = params.data_ptr(), grads.data_ptr()     # (arena slices go to the C ABI as raw addresses: a view costs ~2.5 us)
        self.bn_train = False          # batch statistics instead of running statistics (train-mode BatchNorm)
        # fp8 forward path (0 = off): convolutions with cin >= fp8_min_cin (and cin % 64 == 0) read e4m3 images
        self.fp8_min_cin = int(os.environ.get("DCF_FP8_MIN_CIN", fp8_min_cin))
        # ... and enough output tiles that the launch is MFMA-bound rather than latency-bound: small launches stay on the
        # 16-bit LDS-DMA kernel, which hides the global-load latency better than the fp8 kernel's register staging
        self.fp8_min_blocks = int(os.environ.get("DCF_FP8_MIN_BLOCKS", fp8_min_blocks))
        if self.fp8_min_cin and self.dtype == H.F32:
            raise H.DcfError("the fp8 forward path keeps its activations in bf16 / fp16, not f32")
        self._bn_ws = {}
        self._layout(plan.layers)
        self._sig = None
        self.slabs = None
        self._upload_table(plan.layers)

    # ------------------------------------------------------------------ arenas
    def _layout(self, layers):
        woff = ssoff = 0
        for L in layers:
            K = L.taps * L.cin
            L.wfwd_off = woff
            woff += (L.cout_pad * K * self.es + 255) // 256 * 256
            if L.need_dgrad:
                L.wdgrad_off = woff
                woff += (L.cout_pad * K * self.es + 255) // 256 * 256
            else:
                L.wdgrad_off = -1
            L.shift_off = ssoff
            ssoff += 2 * L.cout_pad
            L.gsum_off = 0
            L.nsplit, L.slab_off = 0, 0
        self.warena = torch.zeros(max(woff, 16), dtype=torch.uint8, device=self.dev)
        self.ssarena = torch.zeros(max(ssoff, 4), dtype=torch.float32, device=self.dev)
        self._wbase, self._ssbase = self.warena.data_ptr(), self.ssarena.data_ptr()
        self.gsum = None
        # fp8 weight images, per-channel dequantisation factors and the activation maxima of the delayed scaling
        w8off = wsoff = 0
        f8 = (H.F8Param * len(layers))()
        for i, L in enumerate(layers):
            L.idx = i
            ok = self.fp8_min_cin > 0 and L.kind != "stem" and L.cin % 64 == 0 and L.cin >= self.fp8_min_cin
            L.w8_off, L.wscale_off = (w8off, wsoff) if ok else (-1, -1)
            if ok:
                w8off += (L.cout_pad * L.taps * L.cin + 255) // 256 * 256
                wsoff += L.cout_pad
            f8[i] = H.F8Param(L.w8_off, L.wscale_off)
        self.has_fp8 = w8off > 0
        if self.has_fp8:
            self.w8arena = torch.zeros(w8off, dtype=torch.uint8, device=self.dev)
            self.wsarena = torch.zeros(wsoff, dtype=torch.float32, device=self.dev)
            self.amax = torch.zeros(len(layers) * H.F8_AMAX_STRIDE, dtype=torch.float32, device=self.dev)
            self.f8table = torch.frombuffer(bytearray(bytes(f8)), dtype=torch.uint8).to(self.dev)
            self.max_cout_pad = max(L.cout_pad for L in layers)

    def _upload_table(self, layers):
        tab = (H.ConvParam * len(layers))()
        for i, L in enumerate(layers):
            bn = (L.gamma_off, L.beta_off, L.mean_off, L.var_off)
            if self.bn_train:           # nothing is folded: plain weights, BN runs as its own kernels
                bn = (-1, -1, -1, -1)
            tab[i] = H.ConvParam(L.w_off, bn[0], bn[1], bn[2], bn[3], L.wfwd_off, L.wdgrad_off,
                                 L.shift_off, L.slab_off, L.gsum_off, L.cout, L.cin, L.taps, L.cout_pad, L.nsplit,
                                 1 if L.kind == "stem" else 0, 0, 0)
        self.table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.dev)
        self.nconv = len(layers)
        self.max_cout = max(L.cout for L in layers)
        self._couts = (ctypes.c_int32 * len(layers))(*[L.cout for L in layers])      # host copy for dcf_wgrad_finalize_rows

    def _w(self, L, dgrad=False):
        # raw device address (the arenas never move): a tensor slice per launch costs the host ~3 us, x160 per step
        return self._wbase + (L.wdgrad_off if dgrad else L.wfwd_off)

    def set_bn_mode(self, train):
        """Returns True when the mode changed (callers drop whatever they captured under the old one)."""
        train = bool(train)
        if train == self.bn_train:
            return False
        self.bn_train = train
        self._upload_table(self.plan.layers)
        self._sig = None
        self.__dict__.pop("_bw_cache", None)      # cached tables describe the other BN mode
        return True

    def _shift(self, L):
        if L.bn is None or self.bn_train:
            return None
        return self._ssbase + 4 * (L.shift_off + L.cout_pad)

    # ------------------------------------------------------------------ step phases
    def check_chains(self):
        """Chain launches bound their waits (a workgroup that gives up records it in its workspace and the results of that launch
        are wrong): once per step the give-up words of every chain workspace are copied to pinned memory asynchronously, and the
        copies of the PREVIOUS step are looked at -- no synchronisation on the step's own path, an error one step late at worst."""
        if not self._chain_ws or torch.cuda.is_current_stream_capturing():
            return
        pend = self.__dict__.get("_chain_pending")
        if pend is not None and pend[1].query():
            bad = pend[0][:pend[2]].nonzero()
            if bad.numel():
                rec = int(pend[0][int(bad[0])])
                self._chain_pending = None
                for ws in self._chain_ws.values():
                    ws[1:2].zero_()                     # reported once: the next steps start clean
                raise H.DcfError("a chain launch (dcf_conv3x3_chain) gave up waiting at layer %d, position tile %d: the results of that "
                                 "step are wrong (another kernel kept part of the launch's workgroups off the GPU for longer than its spin "
                                 "limit?  DCF_CHAIN=0 runs every layer as its own launch)" % ((rec >> 16) & 0x3fff, rec & 0xffff))
            self._chain_pending = pend = None
        if pend is None:
            wss = list(self._chain_ws.values())
            host = self.__dict__.get("_chain_host")
            if host is None or host.numel() < len(wss):
                host = self._chain_host = torch.zeros((max(16, len(wss)),), dtype=torch.int32).pin_memory()
            for i, ws in enumerate(wss):
                host[i:i + 1].copy_(ws[1:2], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._chain_pending = (host, ev, len(wss))

    def prepare(self):
        """Once per step, before forward: fold BN into the compute-dtype weight images."""
        self.check_chains()
        H.call("dcf_weight_prep", self.dtype, self.table, self.nconv, self.params, self.buffers, self.warena, self.ssarena,
               BN_EPS, H.stream_ptr())
        if self.has_fp8 and not self.bn_train:
            H.call("dcf_weight_prep_fp8", self.table, self.f8table, self.nconv, self.max_cout_pad, self.params, self.buffers,
                   self.w8arena, self.wsarena, self.amax, BN_EPS, H.stream_ptr())

    def begin_backward(self, layers):
        sig = tuple(L.out_shape for L in layers)
        if sig != self._sig:
            # one (slab arena, gsum arena, table) per output-shape signature, kept alive: a captured graph has their addresses
            # baked in, and a step stream that alternates between two signatures (valid-point counts) re-uses them
            cache = self.__dict__.setdefault("_bw_cache", {})
            hit = cache.get(sig)
            if hit is None:
                off = goff = 0
                for L in layers:
                    if L.out_shape is None:
                        L.nsplit, L.slab_off, L.gsum_off = 0, 0, 0
                        continue
                    B, Ho, Wo = L.out_shape
                    L.nsplit = ops.conv2d_wgrad_splits(B, Ho, Wo, L.cin, L.cout_pad, L.kh, L.kw, L.stride)
                    L.slab_off = off                        # 256-byte aligned: the 3x3 kernels store 16-byte vectors
                    off += -(-(L.nsplit * L.cout_pad * L.taps * L.cin) // 64) * 64
                    L.gsum_off = goff                       # [4*nsplit][cout_pad] per-wave sums of g (dbeta)
                    goff += 4 * L.nsplit * L.cout_pad
                self.slabs = torch.empty(max(off, 4), dtype=torch.float32, device=self.dev)
                self.gsum = torch.zeros(max(goff, 4), dtype=torch.float32, device=self.dev)
                self._upload_table(layers)
                if len(cache) >= 8:
                    cache.pop(next(iter(cache)))
                cache[sig] = (self.slabs, self.gsum, self.table, [(L.nsplit, L.slab_off, L.gsum_off) for L in layers])
            else:
                self.slabs, self.gsum, self.table, per = hit
                for L, (ns, so, go) in zip(layers, per):
                    L.nsplit, L.slab_off, L.gsum_off = ns, so, go
            self._gsbase, self._slbase = self.gsum.data_ptr(), self.slabs.data_ptr()
            self._sig = sig
        # Only the fusion layers' small parameters (fc1_geo, fc1 / fc2 biases) are ACCUMULATED into the gradient arena (float
        # atomics of the gather backward); every other gradient is written whole by the finalisation launch (eval-mode BN) or by
        # the BatchNorm backward.  They sit at the end of the arena, between the fusion layers' weights: one small fill.
        fus = [L.w_off for L in layers if L.name.startswith("fusion.")]
        if not fus:
            pass
        elif min(fus) > max(L.w_off for L in layers if not L.name.startswith("fusion.")):
            self.grads[min(fus):].zero_()
        else:
            self.grads.zero_()
        self._wq = []                      # weight gradients collected by a backward that did not finish are dropped
        self._gp_pool = None               # the fusion backward's fp32 accumulators: one buffer, one fill per backward
        self.__dict__.pop("_done", None)   # ... and so is its half-finished bucket state (a backward that raised after bucket_ready)

    # Gradient buckets (data parallel): the layer table is ordered LiDAR stream | camera stream | fusion layers, and so is
    # the parameter arena.  With a hook installed (train.Train, world size > 1) the backward finalises FOUR buckets in the
    # order it completes them -- LiDAR stages 4-5 + FPN + heads, the rest of the LiDAR stream + the fusion layers, camera
    # layer4 + FPN, the rest of the camera stream -- and hands each one's arena ranges to the hook, which starts its
    # all-reduce while the backward goes on.  Without a hook: one flush, one finalisation launch (the single-GPU path is
    # unchanged).
    bucket_hook = None

    def _layer_split(self, layers):
        img = [L.idx for L in layers if L.name.startswith("image_")]
        fus = [L.idx for L in layers if L.name.startswith("fusion.")]
        if not img or not fus or max(img) + 1 != min(fus) or max(fus) + 1 != len(layers):
            return None
        return min(img), min(fus)

    def _bucket_layers(self, layers, which):
        """Layer index ranges of a gradient bucket, in the order the backward completes them:
          lidar_hi      LiDAR stages 4-5, FPN, heads  (33 MB of the 50 MB LiDAR stream at cfg2)
          lidar+fusion  the rest of the LiDAR stream + the fusion layers
          image_hi      camera layer4 + camera FPN    (34 MB of the 45 MB camera stream)
        and whatever is left at the end of the backward (camera stem .. layer3)."""
        sp = self._layer_split(layers)
        if sp is None:
            return None
        i0, f0 = sp
        if which == "lidar_hi":
            l4 = [L.idx for L in layers[:i0] if ".layer4." in L.name]
            return [(min(l4), i0)] if l4 else None
        if which == "lidar+fusion":
            return [(0, i0), (f0, len(layers))]
        if which == "image_hi":
            l4 = [L.idx for L in layers[i0:f0] if ".layer4." in L.name]
            return [(min(l4), f0)] if l4 else None
        raise ValueError("unknown gradient bucket %r" % (which,))

    def _finalize(self, lo, hi):
        tab = self.table.data_ptr() + lo * ctypes.sizeof(H.ConvParam)
        H.call("dcf_wgrad_finalize_rows", tab, hi - lo, ctypes.addressof(self._couts) + 4 * lo, self.params, self.buffers, self.ssarena,
               self.slabs, self.gsum, self.grads, BN_EPS, H.stream_ptr())

    def _param_ranges(self, layers, lo, hi):
        """Arena range [a, b) holding every parameter of layers lo..hi-1 (weights, BN affine, fusion biases in between)."""
        a = layers[lo].w_off
        b = layers[hi].w_off if hi < len(layers) else self.params.numel()
        return a, b

    def _finalize_pending(self, layers, ranges):
        """Finalise the not-yet-finalised layers inside the index ranges; returns their arena ranges."""
        done = self.__dict__.setdefault("_done", [False] * len(layers))
        out = []
        for lo, hi in ranges:
            i = lo
            while i < hi:
                if done[i]:
                    i += 1
                    continue
                j = i
                while j < hi and not done[j]:
                    done[j] = True
                    j += 1
                self._finalize(i, j)
                out.append(self._param_ranges(layers, i, j))
                i = j
        return out

    def bucket_ready(self, layers, which):
        """The backward tells that every weight gradient of bucket `which` has been queued: with a hook installed, flush the queue,
        finalise those layers and hand their arena ranges over (their all-reduce starts under the rest of the backward)."""
        if self.bucket_hook is None:
            return
        rng = self._bucket_layers(layers, which)
        if rng is None:
            return
        self._flush_wgrads()
        out = self._finalize_pending(layers, rng)
        if out:
            self.bucket_hook(out)

    def end_backward(self, layers):
        self._flush_wgrads()
        if self.__dict__.get("_done") is None:
            self._finalize(0, self.nconv)
            if self.bucket_hook is not None:
                self.bucket_hook([(0, self.params.numel())])
            return
        out = self._finalize_pending(layers, [(0, len(layers))])
        self.__dict__.pop("_done", None)
        if out:
            self.bucket_hook(out)

    # ------------------------------------------------------------------ convolutions
    def _amax(self, L):
        """(partial maxima [64], previous step's maximum [1]) of the tensor convolution L reads."""
        a = self.amax[L.idx * H.F8_AMAX_STRIDE:]
        return a[:64], a[64:65]

    def _use_fp8(self, L, in_shape):
        """Does convolution L take the fp8 kernel for an input of (B, H, W)?"""
        if not self.has_fp8 or L.w8_off < 0 or self.bn_train:
            return False
        B, Hh, W = in_shape
        M = B * ops.conv_out_size(Hh, L.kh, L.stride, L.pad) * ops.conv_out_size(W, L.kw, L.stride, L.pad)
        return (M + 127) // 128 * (L.cout_pad // 64) >= self.fp8_min_blocks

    def _conv_fwd_fp8(self, L, x, res, relu, nxt):
        f8 = getattr(x, "_f8", None)
        if f8 is None:                       # no producer wrote the image: cast here (and let x's other consumers share it)
            cur, prev = self._amax(L)
            f8 = (ops.cast_fp8(self.dtype, x, prev, cur), prev)
            x._f8 = f8
        want = nxt is not None and self._use_fp8(nxt, y_shape := (x.shape[0], ops.conv_out_size(x.shape[1], L.kh, L.stride, L.pad),
                                                                    ops.conv_out_size(x.shape[2], L.kw, L.stride, L.pad)))
        ncur, nprev = self._amax(nxt) if want else (None, None)
        out = ops.conv2d_fwd_fp8(self.dtype, f8[0], self.w8arena[L.w8_off:], self.wsarena[L.wscale_off:], f8[1], self._shift(L), res,
                                 L.kh, L.kw, L.stride, L.pad, relu, L.cout_pad, want_y8=want, y8amax=nprev, y8cur=ncur)
        if want:
            y, y8 = out
            y._f8 = (y8, nprev)
            return y
        return out

    def conv_fwd(self, L, x, res, relu, nxt=None):
        """nxt: the convolution that will read the result (fp8 path: its e4m3 image is written by this epilogue)."""
        if self.has_fp8 and self._use_fp8(L, x.shape[:3]):
            y = self._conv_fwd_fp8(L, x, res, relu, nxt)
            L.out_shape = (y.shape[0], y.shape[1], y.shape[2])
            return y
        # the step's ~75 plain forward convolutions: per-layer constants cached, arguments marshalled here (ops.conv2d_fwd ->
        # H.call converts 19 generic arguments per launch; this path is ~half its host time)
        shp = x.shape
        c = L.__dict__.get("_fwc")
        if c is None or c[0] != shp:
            B, Hh, W, Cin = shp
            Ho, Wo = ops.conv_out_size(Hh, L.kh, L.stride, L.pad), ops.conv_out_size(W, L.kw, L.stride, L.pad)
            c = L._fwc = (shp, (B, Ho, Wo, L.cout_pad), (B, Hh, W, Cin, Ho, Wo, L.cout_pad, L.kh, L.kw, L.stride, L.pad), (B, Ho, Wo))
        y = torch.empty(c[1], dtype=x.dtype, device=x.device)
        bn = self.bn_train and L.bn is not None          # train-mode BatchNorm: the raw convolution, then the batch-statistics kernels
        rc = self._fn_fwd(self.dtype, x.data_ptr(), self._wbase + L.wfwd_off, self._shift(L), None if (res is None or bn) else res.data_ptr(),
                          y.data_ptr(), *c[2], 1 if (relu and not bn) else 0, H.stream_ptr())
        if rc:
            H.fail("dcf_conv2d_fwd", rc)
        L.out_shape = c[3]
        return self._bn_fwd(L, y, res, relu) if bn else y

    # train-mode BatchNorm as separate kernels (batch statistics; running stats updated in place)
    def _ws(self, C):
        if C not in self._bn_ws:
            self._bn_ws[C] = ops.bn_workspace(C, self.dev)
        return self._bn_ws[C]

    def _bn_fwd(self, L, raw, res, relu):
        # (62 calls per cfg2 step, 62 more in the backward: raw addresses and one tensor for mean | invstd -- the generic
        # ops.bn_train_fwd path built five arena views and three tensors per call)
        C = L.cout_pad
        y = torch.empty_like(raw)
        stat = torch.empty((2, C), dtype=torch.float32, device=raw.device)      # batch mean, 1 / sqrt(var + eps)
        sp = stat.data_ptr()
        rc = self._fn_bnf(self.dtype, raw.data_ptr(), self._pbase + 4 * L.gamma_off, self._pbase + 4 * L.beta_off,
                          None if res is None else res.data_ptr(), y.data_ptr(), sp, sp + 4 * C, self._bbase + 4 * L.mean_off,
                          self._bbase + 4 * L.var_off, raw.numel() // C, C, BN_EPS, 0.1, 1 if relu else 0, self._ws(C).data_ptr(), H.stream_ptr())
        if rc:
            H.fail("dcf_bn_train_fwd", rc)
        L.bn_saved = (raw, stat)
        return y

    def bn_bwd(self, L, g):
        """Gradient through the layer's BatchNorm: identity in eval mode (BN is folded into the conv and its
        chain rule lives in dcf_wgrad_finalize); the batch-statistics backward in train mode."""
        if not (self.bn_train and L.bn is not None):
            return g
        raw, stat = L.bn_saved
        L.bn_saved = None
        C = L.cout_pad
        dx = torch.empty_like(raw)
        sp = stat.data_ptr()
        rc = self._fn_bnb(self.dtype, g.data_ptr(), raw.data_ptr(), sp, sp + 4 * C, self._pbase + 4 * L.gamma_off, self._gbase + 4 * L.gamma_off,
                          self._gbase + 4 * L.beta_off, dx.data_ptr(), raw.numel() // C, C, self._ws(C).data_ptr(), H.stream_ptr())
        if rc:
            H.fail("dcf_bn_train_bwd", rc)
        return dx

    def conv_dgrad(self, L, gy, in_shape, res, mask=None):
        """mask: fused ReLU backward of the layer that produced the tensor gx belongs to."""
        if L.wdgrad_off < 0:
            raise H.DcfError("layer %s was planned without an input gradient" % L.name)
        gshp = gy.shape
        gx = torch.empty(in_shape, dtype=gy.dtype, device=gy.device)
        rc = self._fn_dgrad(self.dtype, gy.data_ptr(), self._wbase + L.wdgrad_off, None if res is None else res.data_ptr(),
                            None if mask is None else mask.data_ptr(), gx.data_ptr(), in_shape[0], in_shape[1], in_shape[2], in_shape[3],
                            gshp[1], gshp[2], gshp[3], L.kh, L.kw, L.stride, L.pad, H.stream_ptr())
        if rc:
            H.fail("dcf_conv2d_dgrad", rc)
        return gx

    # ------------------------------------------------------------------ chains (one launch per residual stage)
    # OFF unless asked for (config `conv_chain: true`, or DCF_CHAIN=1 / force in the environment).  A chain launch needs ALL its
    # workgroups resident at once, which holds when the process has the GPU to itself; two processes sharing one GPU (the
    # two-ranks-on-one-GPU functional test, MPS-style serving) each get part of the CUs and their chain launches starve each other
    # until the bounded spins give up -- found by exactly that test once give-ups were reported (check_chains).  The chains are level
    # with the per-layer launches on time, so the safe form is the default.
    chain_enabled = os.environ.get("DCF_CHAIN", "0") in ("1", "force")

    def can_chain(self, shape, n):
        """Can `n` consecutive 3x3 / stride-1 layers on activations of `shape` [B,H,W,C] run as chain launches
        (dcf_conv3x3_chain: 16-bit storage, eval-mode BatchNorm folded into the weights, no fp8 images, one round of tiles)?"""
        if not self.chain_enabled or n < 2 or self.bn_train or self.has_fp8 or self.dtype == H.F32:
            return False
        # Data-parallel runs (RCCL kernels run beside the backward; functional runs may even put two ranks on one GPU) keep the
        # per-layer launches: a chain launch needs all its workgroups resident at once, and what RCCL's persistent kernels leave
        # free on a CU has never been measured here (no multi-GPU box); the chains are level on time, so
        # nothing is lost.  DCF_CHAIN=force overrides.
        if os.environ.get("DCF_CHAIN") != "force":
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                return False
        key = tuple(shape)
        ok = self._chain_ok.get(key)
        if ok is None:
            B, Hh, W, C = key
            ok = self._chain_ok[key] = ops.conv3x3_chain_supported(self.dtype, B, Hh, W, C, 1)
        return ok

    def _chain(self, x, layers, flip):
        """layers: (weight address, shift address, res, mask, relu) per layer, res / mask = tensor, None or the index of an
        earlier layer of this call.  Chains longer than the ABI's limit are split (an index that then points into an earlier
        launch becomes that launch's tensor)."""
        B, Hh, W, C = x.shape
        key = (B, Hh, W, C)
        ws = self._chain_ws.get(key)
        if ws is None:
            ws = self._chain_ws[key] = ops.conv3x3_chain_workspace(self.dtype, B, Hh, W, C, H.CHAIN_MAX_LAYERS, x.device)
        outs, cur, pos = [], x, 0
        while pos < len(layers):
            part = layers[pos:pos + H.CHAIN_MAX_LAYERS]
            fix = []
            for (w, sh, r, m, relu) in part:
                r = (outs[r] if r < pos else r - pos) if type(r) is int else r
                m = (outs[m] if m < pos else m - pos) if type(m) is int else m
                fix.append((w, sh, r, m, relu))
            tab = self._chain_tab.get(len(fix))
            if tab is None:
                tab = self._chain_tab[len(fix)] = (H.ChainLayer * len(fix))()
            got = ops.conv3x3_chain(self.dtype, cur, fix, flip, ws, tab)
            outs.extend(got)
            cur = got[-1]
            pos += len(part)
        return outs

    def chain_fwd(self, x, specs):
        """specs: (layer, res, relu) per layer, each layer reading the previous one's output (the first reads x).  Same results
        as conv_fwd layer by layer.  Returns the list of outputs."""
        outs = self._chain(x, [(self._wbase + L.wfwd_off, self._shift(L), r, None, relu) for (L, r, relu) in specs], 0)
        shp = (x.shape[0], x.shape[1], x.shape[2])
        for (L, _, _) in specs:
            L.out_shape = shp
        return outs

    def chain_dgrad(self, g, specs):
        """specs: (layer, res, mask) per layer = conv_dgrad(layer, previous output, shape, res, mask) layer by layer."""
        for (L, _, _) in specs:
            if L.wdgrad_off < 0:
                raise H.DcfError("layer %s was planned without an input gradient" % L.name)
        return self._chain(g, [(self._wbase + L.wdgrad_off, None, r, m, False) for (L, r, m) in specs], 1)

    def shortcut_dgrad(self, Ld, dd, L1, g1, in_shape, mask=None):
        """Input gradient of a strided block whose shortcut is a 1x1 / stride-2 conv Ld and whose main path starts with the
        3x3 / stride-2 conv L1: the shortcut's gradient touches only the even pixels of x, so it is computed as a dense 1x1 GEMM
        on dd's own grid (a quarter of the pixels, no zero stores) and joins L1's input gradient in that launch's epilogue.
        Values are those of dgrad(L1, g1, res=dgrad(Ld, dd)): the shortcut's term is rounded to the compute dtype first in both."""
        if L1.wdgrad_off < 0 or Ld.wdgrad_off < 0:
            raise H.DcfError("layer %s was planned without an input gradient" % L1.name)
        B, Hq, Wq, _ = dd.shape
        gq = ops.conv2d_dgrad(self.dtype, dd, self._w(Ld, True), None, (B, Hq, Wq, in_shape[3]), 1, 1, 1, 0)
        return ops.conv2d_dgrad_halfres(self.dtype, g1, self._w(L1, True), None, gq, in_shape, L1.kh, L1.kw, L1.stride, L1.pad, mask)

    def _gs(self, L):
        return self._gsbase + 4 * L.gsum_off if (L.bn is not None and not self.bn_train) else None

    def _flush_wgrads(self):
        """Issue the collected weight gradients together (dcf_conv2d_wgrad_group: one launch per <= 32 layers of a kernel class)."""
        q = self._wq
        if not q:
            return
        # the table of a step is the previous step's with new x / gy addresses: the ctypes array is kept per layer sequence
        # (80 sixteen-field constructors per step were 0.3 ms of host time)
        key = tuple([t[0].idx for t in q])
        ent = self._wtab.get(key)
        shapes = [t[1].shape for t in q]
        sig = (self.bn_train, self._slbase, self._gsbase, tuple([(t[0].nsplit, t[0].slab_off, t[0].gsum_off) for t in q]))
        if ent is None or ent[1] != shapes or ent[2] != sig:
            items = (H.WgradItem * len(q))()
            for i, (L, x, gy) in enumerate(q):
                B, Hh, W, Cin = x.shape
                items[i] = H.WgradItem(self.dtype, L.nsplit, 0, 0, self._slbase + 4 * L.slab_off, self._gs(L), B, Hh, W, Cin, L.cout_pad, L.kh, L.kw, L.stride, L.pad, 0)
            if len(self._wtab) > 16:
                self._wtab.clear()
            ent = self._wtab[key] = (items, shapes, sig)
        items = ent[0]
        for i, (L, x, gy) in enumerate(q):
            it = items[i]
            it.x = x.data_ptr()
            it.gy = gy.data_ptr()
        H.call("dcf_conv2d_wgrad_group", ctypes.addressof(items), len(q), H.stream_ptr())
        self._wq = []                      # the launches are enqueued: x / gy may be released (same stream)

    _wq = []
    _fus_ws = None
    _gp_pool = None
    _group = os.environ.get("DCF_WGRAD_GROUP", "1") != "0"
    # one-writer-per-point fusion backward (dcf_fusion_gather_bwd_pts: no zero-fill, no atomics on dP, no cast): correct, but a
    # point that thousands of pixels chose is then one wave's serial work -- measured 1.07 vs 0.30 ms per step at cfg2, so off
    _fusion_pts = os.environ.get("DCF_FUSION_PTS", "0") == "1"
    # "0" = the fp32 accumulator + cast form of the inverse-map backward again (A/B runs)
    _fusion_direct = os.environ.get("DCF_FUSION_DIRECT", "1") != "0"

    def conv_wgrad(self, L, x, gy, defer=True):
        """defer=False: gy (or x) is modified in place later in the backward (e.g. masked by a ReLU) -- launch now."""
        if self._group and defer and L.kind != "stem":
            # independent of everything else in the backward: collected (x, gy kept alive) and issued together at the end
            q = self.__dict__.get("_wq")
            if q is None:
                q = self._wq = []
            q.append((L, x, gy))
            return
        ops.conv2d_wgrad(self.dtype, x, gy, self._slbase + 4 * L.slab_off, L.nsplit, L.kh, L.kw, L.stride, L.pad, self._gs(L))

    def stem_fwd(self, L, img4, Hh, W):
        if self.bn_train:
            raw = ops.stem7x7_fwd(self.dtype, img4, self._w(L), None, False, L.cout_pad, Hh, W)
            L.out_shape = (raw.shape[0], raw.shape[1], raw.shape[2])
            return self._bn_fwd(L, raw, None, True)
        y = ops.stem7x7_fwd(self.dtype, img4, self._w(L), self._shift(L), True, L.cout_pad, Hh, W)
        L.out_shape = (y.shape[0], y.shape[1], y.shape[2])
        return y

    def stem_wgrad(self, L, img4, gy, Hh, W):
        ops.stem7x7_wgrad(self.dtype, img4, gy, self._slbase + 4 * L.slab_off, L.nsplit, Hh, W, self._gs(L))

    def relu_mask(self, g, y):
        """g *= (y > 0) in place (the dbeta sums come out of the wgrad kernel)."""
        ops.relu_bwd_chansum(self.dtype, g, y, None, True)
        return g

    def wait_event(self, ev):
        """Make the compute stream wait for work recorded on another stream (side-stream geometry)."""
        torch.cuda.current_stream().wait_event(ev)

    # ------------------------------------------------------------------ elementwise
    def to_device(self, t):
        return t.to(self.dev)

    def nchw_to_nhwc(self, x):
        if x.dtype != torch.float32:
            raise H.DcfError("x_lidar must be float32 [B,Cz,L,W] (model.py:194)")
        return ops.nchw_to_nhwc(x.contiguous(), self.dtype)

    def nhwc_to_nchw(self, x):
        return ops.nhwc_to_nchw(x, self.dtype)

    def image_to_nhwc4(self, img):
        if img.dtype != torch.uint8:
            raise H.DcfError("x_image must be uint8 [B,3,H,W] (data_import_carla.py:62)")
        return ops.image_to_nhwc4(img.contiguous(), self.dtype)

    def resize_fwd(self, x, out_hw, align, add):
        return ops.resize_bilinear_fwd(self.dtype, x, out_hw, align, add)

    def resize_bwd(self, gy, in_hw, align):
        return ops.resize_bilinear_bwd(self.dtype, gy, in_hw, align)

    def maxpool_fwd(self, x):
        return ops.maxpool_fwd(self.dtype, x)

    def maxpool_fwd_idx(self, x):
        return ops.maxpool_fwd_idx(self.dtype, x)

    def maxpool_bwd(self, x, gy, y=None, idx=None):
        if idx is not None:
            return ops.maxpool_bwd_idx(self.dtype, idx, gy, tuple(x.shape))
        return ops.maxpool_bwd(self.dtype, x, y, gy)

    def head_fwd(self, head, anchors):
        return ops.head_fwd(self.dtype, head, anchors)

    def head_bwd(self, head, anchors, pred, gpred):
        return ops.head_bwd(self.dtype, head, anchors, pred, gpred.contiguous())

    def cast_like(self, src, like):
        if src.dtype == like.dtype:
            return src
        return ops.cast(src, self.dtype)

    # ------------------------------------------------------------------ fusion
    def point_sample_fwd(self, fmap, uv, cnt, n_max):
        B = fmap.shape[0]
        # frames side by side; the kernel writes every row (zeros past a frame's point count): no fill
        fp = (torch.empty if n_max > 0 else torch.zeros)((B, max(n_max, 1), fmap.shape[-1]), dtype=fmap.dtype, device=fmap.device)
        if uv.is_contiguous() and cnt.is_contiguous() and n_max > 0:
            return ops.point_sample_fwd_batch(self.dtype, fmap, uv, cnt, n_max, fp)               # one launch for the batch
        for b in range(B):
            ops.point_sample_fwd(self.dtype, fmap[b], uv[b], cnt[b:b + 1], n_max, out=fp[b])
        return fp

    def point_sample_bwd(self, gfp, uv, cnt, n_max, fmap_shape, gF):
        if gF is None:
            gF = torch.zeros(fmap_shape, dtype=torch.float32, device=self.dev)
        if uv.is_contiguous() and cnt.is_contiguous() and gfp.is_contiguous() and n_max > 0 and gfp.shape[1] == n_max:
            return ops.point_sample_bwd_batch(self.dtype, gfp, uv, cnt, n_max, gF)
        for b in range(gfp.shape[0]):
            ops.point_sample_bwd(self.dtype, gfp[b], uv[b], cnt[b:b + 1], n_max, gF[b])
        return gF

    def fusion_gather_fwd(self, P, xyz, idx, stride, aff, w1d_off, b1_off):
        B, (K_, h, w) = P.shape[0], idx.shape[-3:]
        hsum = torch.empty((B, h, w, P.shape[2]), dtype=P.dtype, device=P.device)
        cnt = torch.empty((B, h * w), dtype=torch.float32, device=P.device)
        if P.is_contiguous() and xyz.is_contiguous() and idx.is_contiguous():
            return ops.fusion_gather_fwd_batch(self.dtype, P, xyz, idx, stride, aff, self._pbase + 4 * w1d_off, self._pbase + 4 * b1_off, hsum, cnt)
        for b in range(B):
            ops.fusion_gather_fwd(self.dtype, P[b], xyz[b], idx[b], stride, aff, self._pbase + 4 * w1d_off, self._pbase + 4 * b1_off, out=(hsum[b], cnt[b]))
        return hsum, cnt

    def fusion_gather_bwd(self, P, xyz, idx, stride, aff, w1d_off, b1_off, ghsum, inv=None, site=0, inv_nmax=None):
        """inv: inverse KNN maps of the step (ops.fusion_invert, map = site*B + frame) -> the point-sorted backward;
        else the pixel-run one."""
        use_inv = inv is not None and P.shape[2] % 64 == 0 and 64 <= P.shape[2] <= 256
        if use_inv and self._fusion_pts:
            # one writer per point row: the gradient comes out whole, in the compute dtype (no zero-fill, no atomics on it, no cast)
            gP = torch.empty(P.shape, dtype=P.dtype, device=self.dev)
            for b in range(P.shape[0]):
                ops.fusion_gather_bwd_pts(self.dtype, P[b], xyz[b], inv, inv_nmax or P.shape[1], site * P.shape[0] + b, tuple(idx.shape[-3:]), stride, aff,
                                          self._pbase + 4 * w1d_off, self._pbase + 4 * b1_off, ghsum[b], gP[b], self._gbase + 4 * w1d_off, self._gbase + 4 * b1_off)
            return gP
        batched = use_inv and P.is_contiguous() and xyz.is_contiguous() and ghsum.is_contiguous() and P.shape[0] <= 64
        if use_inv and self._fus_ws is None:
            self._fus_ws = ops.fusion_bwd_workspace(self.dev)             # zeroed once: the kernel leaves it zero
        if batched and self._fusion_direct:
            # gP in the compute dtype, one (half-size) fill for the four sites, no cast: see dcf_fusion_gather_bwd_direct_batch
            gP = self._gp_zeros(P.shape, P.dtype)
            me = idx.shape[-3] * idx.shape[-2] * idx.shape[-1]
            key = (me, P.shape[2], P.shape[0])
            dws = self._fus_dws.get(key)
            if dws is None:
                dws = self._fus_dws[key] = ops.fusion_bwd_direct_workspace(self.dev, *key)
            ops.fusion_gather_bwd_direct_batch(self.dtype, P, xyz, inv, inv_nmax or P.shape[1], site * P.shape[0], tuple(idx.shape[-3:]), stride, aff,
                                               self._pbase + 4 * w1d_off, self._pbase + 4 * b1_off, ghsum, gP, self._gbase + 4 * w1d_off, self._gbase + 4 * b1_off,
                                               self._fus_ws, dws)
            return gP
        gP = self._gp_zeros(P.shape)
        if use_inv:
            if batched:
                ops.fusion_gather_bwd_inv_batch(self.dtype, P, xyz, inv, inv_nmax or P.shape[1], site * P.shape[0], tuple(idx.shape[-3:]), stride, aff,
                                                self._pbase + 4 * w1d_off, self._pbase + 4 * b1_off, ghsum, gP, self._gbase + 4 * w1d_off, self._gbase + 4 * b1_off, self._fus_ws)
                return gP
        for b in range(P.shape[0]):
            if use_inv:
                ops.fusion_gather_bwd_inv(self.dtype, P[b], xyz[b], inv, inv_nmax or P.shape[1], site * P.shape[0] + b, tuple(idx.shape[-3:]), stride, aff, self.params[w1d_off:],
                                          self.params[b1_off:], ghsum[b], gP[b], self._gbase + 4 * w1d_off, self._gbase + 4 * b1_off, self._fus_ws)
            else:
                ops.fusion_gather_bwd(self.dtype, P[b], xyz[b], idx[b], stride, aff, self._pbase + 4 * w1d_off, self._pbase + 4 * b1_off,
                                      ghsum[b], gP[b], self._gbase + 4 * w1d_off, self._gbase + 4 * b1_off)
        return gP

    def _gp_zeros(self, shape, dtype=torch.float32):
        """Zeroed [B, rows, Cb] gradient / accumulator of a site's fusion backward.  The sites of one backward (same B and rows) are
        carved out of ONE buffer zeroed by one fill when the first of them asks: four fills per step become one."""
        B, rows, cb = shape
        pool = self._gp_pool
        need = B * rows * cb
        if pool is None or pool[1] != (B, rows, dtype) or pool[2] + need > pool[0].numel():
            total = B * rows * sum(f["cb"] for f in self.plan.fusion) if getattr(self.plan, "fusion", None) else need
            pool = self._gp_pool = [torch.zeros(max(total, need), dtype=dtype, device=self.dev), (B, rows, dtype), 0]
        off = pool[2]
        pool[2] = off + need
        return pool[0][off:off + need].view(B, rows, cb)

    def conv_fwd_rowscale(self, L, x, res, cnt, b2_off):
        """conv_fwd of a 1x1 layer without BatchNorm + cnt[m] * b2[c] in the same epilogue (the fusion site's fc2 under the neighbour
        sum); the fp8 kernel has no such epilogue: there the bias stays a pass of its own."""
        if self._use_fp8(L, x.shape[:3]) or L.bn is not None or L.kh != 1 or L.kw != 1 or L.stride != 1 or L.cout_pad != L.cout:
            # (a padded layer: the epilogue would read cout_pad entries of the cout-long master bias)
            return self.rowscale_bias_fwd(self.conv_fwd(L, x, res, False), cnt, b2_off)
        y = ops.conv2d_fwd_rowscale(self.dtype, x, self._w(L), self._pbase + 4 * b2_off, cnt, res, False, L.cout_pad)
        L.out_shape = (y.shape[0], y.shape[1], y.shape[2])
        return y

    def rowscale_bias_fwd(self, y, cnt, b2_off):
        return ops.rowscale_bias_fwd(self.dtype, y, cnt, self.params[b2_off:])

    def relu_mask_rowscale_bwd(self, gy, y, cnt, b2_off):
        """At a fusion site: the masked gradient for the stage's last block (a new tensor) + fc2's bias gradient, one pass."""
        return ops.relu_mask_rowscale_bwd(self.dtype, gy, y, cnt, self._gbase + 4 * b2_off)

    def rowscale_bias_bwd(self, gy, cnt, b2_off):
        ops.rowscale_bias_bwd(self.dtype, gy, cnt, self._gbase + 4 * b2_off)
