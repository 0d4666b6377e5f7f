"""Static execution plan of the continuous-fusion network (forward + hand-written backward).

The reference builds the network as an nn.Module tree and lets autograd + cuDNN run it
(/root/reference/model.py:10-204).  Here the network is a fixed list of layer descriptors
over flat arenas, executed by explicit kernel calls:

  * parameters live in ONE flat fp32 arena (convolution weights physically [O][kh][kw][I]),
    gradients in a second arena with the same offsets, Adam moments in two more -- one
    all-reduce and one optimiser launch per step;
  * eval-mode BatchNorm (what the reference really trains with) is folded into
    the convolution: y = act(conv(x, scale*W) + shift + residual) is one kernel, activations
    are written once and read once;
  * the backward pass is spelled out per block; the folded-BN chain rule never needs the
    pre-BN tensor:  with G = wgrad(g, x) on the UNSCALED weights,
        dW = scale*G,  dbeta = sum g,  dgamma = (<W,G> - mean*dbeta) * rsqrt(var+eps).

All arithmetic goes through a backend object (`backend_hip.HipBackend`: HIP kernels through
the C ABI).  The engine itself only sequences calls; tests may pass a different backend to
check the sequencing/maths on CPU, the product never does.
"""
import os

import numpy as np
import torch

# experiments (read once per process): "0" = no inverse KNN maps at all, "2" = build them but run the pixel-run backward
FUSION_INV = os.environ.get("DCF_FUSION_INV", "1")
# fusion sites: fc2's bias gradient and the ReLU mask of the stage's last block in one pass (0 = the two passes of rounds 1-5)
FUSED_SITE_MASK = os.environ.get("DCF_FUSED_SITE_MASK", "1") != "0"
# "0" = the 1x1 / stride-2 shortcut's input gradient as a full-resolution tensor again (A/B runs; same values)
HALFRES_SHORTCUT = os.environ.get("DCF_HALFRES_SHORTCUT", "1") != "0"


class ConvLayer(object):
    """One convolution (or linear = 1x1) of the plan = one row of the dcf_conv_param table."""

    def __init__(self, idx, name, cin, cout, kh, kw, stride, pad, bn=None, kind="conv", need_dgrad=True, names=None):
        self.idx, self.name, self.bn, self.kind = idx, name, bn, kind
        self.cin, self.cout, self.kh, self.kw, self.stride, self.pad = cin, cout, kh, kw, stride, pad
        self.cout_pad = (cout + 31) // 32 * 32
        self.need_dgrad = need_dgrad
        self.names = names or [name]          # parameter keys making up the weight (heads: two)
        self.w_off = self.gamma_off = self.beta_off = self.mean_off = self.var_off = -1
        self.out_shape = None                  # (B, Ho, Wo) of the last forward

    @property
    def taps(self):
        return self.kh * self.kw


class ParamTable(object):
    """Flat parameter / buffer arenas and the state_dict views into them."""

    def __init__(self):
        self.entries = []      # (key, logical_shape, offset, numel, layout)
        self.buffers = []      # (key, shape, offset, numel)
        self.n_params = 0
        self.n_buffers = 0

    def add_param(self, key, shape, layout="plain"):
        n = int(np.prod(ParamTable.physical_shape(shape, layout)))
        off = self.n_params
        self.entries.append((key, tuple(shape), off, n, layout))
        self.n_params += (n + 3) // 4 * 4     # keep every tensor 16-byte aligned
        return off

    def add_buffer(self, key, shape):
        n = int(np.prod(shape)) if len(shape) else 1
        off = self.n_buffers
        self.buffers.append((key, tuple(shape), off, n))
        self.n_buffers += (n + 3) // 4 * 4
        return off

    @staticmethod
    def physical_shape(shape, layout):
        if layout == "ohwi":                   # logical [O,I,kh,kw] stored [O,kh,kw,I]
            O, I, kh, kw = shape
            return (O, kh, kw, I)
        if layout == "stem":                   # logical [O,3,7,7] stored [O,7,8,4] (zero padded)
            return (shape[0], 7, 8, 4)
        return tuple(shape)

    @staticmethod
    def view(flat, shape, off, n, layout):
        """Logical-shape view of a parameter inside the flat arena."""
        seg = flat[off:off + n]
        if layout == "ohwi":
            O, I, kh, kw = shape
            return seg.view(O, kh, kw, I).permute(0, 3, 1, 2)
        if layout == "stem":
            return seg.view(shape[0], 7, 8, 4)[:, :, :7, :3].permute(0, 3, 1, 2)
        return seg.view(shape)


class Block(object):
    """Residual block: relu(bn2(conv2(relu(bn1(conv1 x)))) + shortcut(x))  (model.py:32-41)."""

    def __init__(self, conv1, conv2, down=None):
        self.conv1, self.conv2, self.down = conv1, conv2, down
        self.saved = None
        self.next_conv = None          # conv1 of the block that consumes this block's output (same stage), if any

    def forward(self, K, x, save=True):
        r = K.conv_fwd(self.down, x, None, False) if self.down is not None else x
        y1 = K.conv_fwd(self.conv1, x, None, True, self.conv2)
        y = K.conv_fwd(self.conv2, y1, r, True, self.next_conv)
        x._f8 = y1._f8 = None          # fp8 images (fp8 path) are dead once their consumers ran
        self.saved = (x, y1, y) if save else None
        return y

    def backward(self, K, g, extra=None, need_gx=True, g_masked=False, prev=None):
        """g = dL/dy (consumed / overwritten).  extra = gradient reaching x from other consumers.
        g_masked: the producer of g already applied this block's output ReLU mask.
        prev: the Block whose output is this block's input x -- its ReLU backward (mask by x) is then
        fused into the epilogue of the dgrad that produces gx."""
        x, y1, y = self.saved
        self.saved = None
        g2 = g if g_masked else K.relu_mask(g, y)
        d2 = K.bn_bwd(self.conv2, g2)                                                 # identity with folded (eval) BN
        K.conv_wgrad(self.conv2, y1, d2)                                              # also dbeta(bn2) = sum g2
        g1 = K.bn_bwd(self.conv1, K.conv_dgrad(self.conv2, d2, tuple(y1.shape), None, y1))   # fused ReLU mask of y1
        return self.backward_tail(K, x, g2, g1, extra, need_gx, prev)

    def backward_tail(self, K, x, g2, g1, extra, need_gx, prev):
        """The rest of backward() from g1 = dL/d(conv1 output, masked) on: conv1's weight gradient, the shortcut, dL/dx."""
        K.conv_wgrad(self.conv1, x, g1)
        pm = x if prev is not None else None
        if self.down is not None:
            dd = K.bn_bwd(self.down, g2)
            K.conv_wgrad(self.down, x, dd)
            if not need_gx:
                return None
            d, c = self.down, self.conv1
            if (extra is None and HALFRES_SHORTCUT and d.kh == 1 and d.kw == 1 and d.stride == 2 and d.pad == 0
                    and c.kh == 3 and c.stride == 2 and c.pad == 1):
                return K.shortcut_dgrad(d, dd, c, g1, tuple(x.shape), pm)
            gx = K.conv_dgrad(self.down, dd, tuple(x.shape), extra)
            return K.conv_dgrad(self.conv1, g1, tuple(x.shape), gx, pm)
        if not need_gx:
            return None
        assert extra is None, "an identity-shortcut block cannot take an extra gradient"
        return K.conv_dgrad(self.conv1, g1, tuple(x.shape), g2, pm)


def blocks_forward(K, blocks, x, save=True):
    """The blocks of one residual stage, in order (model.py:48-60).  Where the backend can (16-bit storage, folded eval-mode
    BatchNorm, a layer's tiles in one round of workgroups: HipBackend.can_chain), every 3x3 / stride-1 convolution of the
    stage behind its first strided one -- block 0's conv2, then conv1 / conv2 of every following block -- runs as ONE chain
    launch (dcf_conv3x3_chain) instead of one launch each: same kernels, same results, no kernel boundary between layers."""
    n = len(blocks)
    ok = (n >= 1 and getattr(K, "can_chain", None) is not None and all(type(b) is Block for b in blocks)
          and all(b.down is None and b.conv1.stride == 1 for b in blocks[1:]))
    if ok:
        b0 = blocks[0]
        strided = b0.down is not None or b0.conv1.stride != 1 or b0.conv1.cin != b0.conv1.cout
        nchain = 2 * n - 1 if strided else 2 * n
        B, Hh, W, _ = x.shape
        s0 = b0.conv1.stride
        shape = (B, (Hh + 2 - 3) // s0 + 1, (W + 2 - 3) // s0 + 1, b0.conv1.cout) if strided else tuple(x.shape)
        ok = nchain >= 2 and all(L.kh == 3 and L.kw == 3 and L.pad == 1 and L.cout == L.cout_pad == shape[3]
                                 for b in blocks for L in (b.conv1, b.conv2)) and K.can_chain(shape, nchain)
    if not ok:
        for b in blocks:
            x = b.forward(K, x, save)
        return x
    specs, ins = [], []                   # per chain layer: (layer, residual, relu); ins[k] = block k's input (tensor or chain index)
    if strided:
        r = K.conv_fwd(b0.down, x, None, False) if b0.down is not None else x
        y1 = K.conv_fwd(b0.conv1, x, None, True)
        x0 = y1                            # the chain's input
        specs.append((b0.conv2, r, True))
        first = 1
    else:
        x0 = x
        first = 0
    xin = len(specs) - 1                   # chain index of the current block's input (-1: the chain's own input, only when not strided)
    for b in blocks[first:]:
        specs.append((b.conv1, None, True))
        specs.append((b.conv2, xin if xin >= 0 else x, True))
        xin = len(specs) - 1
    outs = K.chain_fwd(x0, specs)
    if save:
        k = 0
        if strided:
            b0.saved = (x, y1, outs[0])
            k = 1
        prev = outs[0] if strided else x
        for b in blocks[first:]:
            b.saved = (prev, outs[k], outs[k + 1])
            prev = outs[k + 1]
            k += 2
    else:
        for b in blocks:
            b.saved = None
    return outs[-1]


def blocks_backward(K, blocks, g, masked, extra0=None, prev0=None, need_gx0=True):
    """Backward of one stage's blocks, last block first (what the callers' loops over Block.backward did): g = dL/d(stage output),
    masked = its producer already applied the last block's ReLU mask, extra0 / prev0 / need_gx0 = block 0's `extra`, `prev`
    and `need_gx`.  Returns (dL/d(stage input) or None, whether it is already masked by prev0's output).  With a chain-capable
    backend the input-gradient convolutions of the stage -- dgrad(conv2) and dgrad(conv1) of every identity block, last block
    first, and block 0's dgrad(conv2) -- are one chain launch; the weight gradients are queued as before."""
    n = len(blocks)
    ok = (n >= 1 and getattr(K, "can_chain", None) is not None and all(type(b) is Block and b.saved is not None for b in blocks)
          and all(b.down is None and b.conv1.stride == 1 for b in blocks[1:]))
    if ok:
        b0 = blocks[0]
        strided = b0.down is not None or b0.conv1.stride != 1 or b0.conv1.cin != b0.conv1.cout
        ident0 = not strided and need_gx0 and extra0 is None       # block 0's dgrad(conv1) joins the chain too
        nchain = 2 * (n - 1) + 1 + (1 if ident0 else 0)
        ok = nchain >= 2 and all(L.kh == 3 and L.kw == 3 and L.pad == 1 and L.cout == L.cout_pad == g.shape[3]
                                 for b in blocks for L in (b.conv1, b.conv2)) and K.can_chain(tuple(g.shape), nchain)
    if not ok:
        for bi in range(n - 1, -1, -1):
            prev = blocks[bi - 1] if bi > 0 else prev0
            g = blocks[bi].backward(K, g, extra0 if bi == 0 else None, need_gx=(need_gx0 if bi == 0 else True), g_masked=masked, prev=prev)
            masked = prev is not None
        return g, masked
    g2 = g if masked else K.relu_mask(g, blocks[-1].saved[2])
    specs, wq = [], []                     # chain layers (layer, residual, mask); queued weight gradients (layer, x, gy) with gy a tensor or a chain index
    cur = g2                               # dL/d(block output), masked: tensor (the chain's input) or chain index
    for bi in range(n - 1, 0, -1):
        b = blocks[bi]
        x, y1, _ = b.saved
        b.saved = None
        specs.append((b.conv2, None, y1))                          # g1 = dgrad(conv2, g2) * (y1 > 0)
        wq.append((b.conv2, y1, cur))
        ig1 = len(specs) - 1
        specs.append((b.conv1, cur, x))                            # gx = (dgrad(conv1, g1) + g2) * (x > 0): x is blocks[bi - 1]'s output
        wq.append((b.conv1, x, ig1))
        cur = len(specs) - 1
    x, y1, _ = b0.saved
    b0.saved = None
    specs.append((b0.conv2, None, y1))
    wq.append((b0.conv2, y1, cur))
    ig1 = len(specs) - 1
    if ident0:
        specs.append((b0.conv1, cur, x if prev0 is not None else None))
    outs = K.chain_dgrad(g2, specs)
    for (L, xx, gy) in wq:
        K.conv_wgrad(L, xx, outs[gy] if type(gy) is int else gy)
    g2_0 = outs[cur] if type(cur) is int else cur
    if ident0:
        K.conv_wgrad(b0.conv1, x, outs[ig1])
        return outs[-1], prev0 is not None
    gx = b0.backward_tail(K, x, g2_0, outs[ig1], extra0, need_gx0, prev0)
    return gx, prev0 is not None


class Bottleneck(object):
    """ResNet-50 block (torchvision v1.5: the stride sits on the 3x3):
    relu(bn3(conv3_1x1(relu(bn2(conv2_3x3_s(relu(bn1(conv1_1x1 x))))))) + shortcut(x)).  Same calling convention as Block."""

    def __init__(self, conv1, conv2, conv3, down=None):
        self.conv1, self.conv2, self.conv3, self.down = conv1, conv2, conv3, down
        self.saved = None
        self.next_conv = None

    def forward(self, K, x, save=True):
        r = K.conv_fwd(self.down, x, None, False) if self.down is not None else x
        y1 = K.conv_fwd(self.conv1, x, None, True, self.conv2)
        y2 = K.conv_fwd(self.conv2, y1, None, True, self.conv3)
        y = K.conv_fwd(self.conv3, y2, r, True, self.next_conv)
        x._f8 = y1._f8 = y2._f8 = None
        self.saved = (x, y1, y2, y) if save else None
        return y

    def backward(self, K, g, extra=None, need_gx=True, g_masked=False, prev=None):
        x, y1, y2, y = self.saved
        self.saved = None
        g3 = g if g_masked else K.relu_mask(g, y)
        d3 = K.bn_bwd(self.conv3, g3)
        K.conv_wgrad(self.conv3, y2, d3)
        g2 = K.bn_bwd(self.conv2, K.conv_dgrad(self.conv3, d3, tuple(y2.shape), None, y2))   # fused ReLU mask of y2
        K.conv_wgrad(self.conv2, y1, g2)
        g1 = K.bn_bwd(self.conv1, K.conv_dgrad(self.conv2, g2, tuple(y1.shape), None, y1))   # fused ReLU mask of y1
        K.conv_wgrad(self.conv1, x, g1)
        pm = x if prev is not None else None
        if self.down is not None:
            dd = K.bn_bwd(self.down, g3)
            K.conv_wgrad(self.down, x, dd)
            if not need_gx:
                return None
            gx = K.conv_dgrad(self.down, dd, tuple(x.shape), extra)
            return K.conv_dgrad(self.conv1, g1, tuple(x.shape), gx, pm)
        if not need_gx:
            return None
        assert extra is None, "an identity-shortcut block cannot take an extra gradient"
        return K.conv_dgrad(self.conv1, g1, tuple(x.shape), g3, pm)


class StackPlan(object):
    """Residual stages on their own: the execution plan behind the reference's ResidualBlock / ResidualBlockModule /
    ResnetCustomed module surfaces (model.py:10-79).  stages = list (stage) of lists (block) of (key prefix, cin, cout);
    taps = indices of the stages whose outputs are returned, in return order.  Input and outputs are NCHW fp32 like the
    reference's; inside, the same Block objects and kernels as the full network."""

    def __init__(self, stages, taps):
        self.table = ParamTable()
        self.layers = []
        self.stages = []
        self.taps = list(taps)
        for blocks in stages:
            bl = []
            for pfx, ci, co in blocks:
                if ci % 32 or co % 32:
                    raise ValueError("channel counts must be multiples of 32 (MFMA 32x32 tiles)")
                s = 2 if ci != co else 1                               # model.py:14-19, :43-45
                c1 = self._conv(pfx + "conv1", ci, co, (3, 3), s, pfx + "bn1")
                c2 = self._conv(pfx + "conv2", co, co, (3, 3), 1, pfx + "bn2")
                dn = self._conv(pfx + "down_conv", ci, co, (1, 1), 2, pfx + "down_bn") if ci != co else None
                bl.append(Block(c1, c2, dn))
            for a_, b_ in zip(bl[:-1], bl[1:]):
                a_.next_conv = b_.conv1
            self.stages.append(bl)
        for t in self.taps:
            if t != len(self.stages) - 1 and self.stages[t + 1][0].down is None:
                raise NotImplementedError("an intermediate output joins the backward through the next stage's shortcut convolution")
        self.ctx = None

    def forward(self, K, x_nchw, save=True):
        x = K.nchw_to_nhwc(x_nchw)
        outs = {}
        for si, blocks in enumerate(self.stages):
            x = blocks_forward(K, blocks, x, save)
            outs[si] = x
        self.ctx = True if save else None
        return [K.nhwc_to_nchw(outs[t]) for t in self.taps]

    def backward(self, K, gouts):
        """gouts: gradients of the returned outputs (NCHW fp32, None = no gradient).  Returns dL/dx NCHW fp32."""
        if not self.ctx:
            raise RuntimeError("backward through a forward that ran without saving activations")
        self.ctx = None
        have = [t for t, g in zip(self.taps, gouts) if g is not None]
        top = max(have) if have else -1
        # stages past the last output that carries a gradient take no part in this backward: their layers are planned with no
        # slabs (out_shape None -> nsplit 0), so that the finalisation launch writes ZERO gradients for them instead of reducing
        # whatever an earlier backward left in the slab arena
        for si in range(top + 1, len(self.stages)):
            for b in self.stages[si]:
                b.saved = None
                for L in (b.conv1, b.conv2, b.down):
                    if L is not None:
                        L.out_shape = None
        K.begin_backward(self.layers)
        gmap = dict((t, K.nchw_to_nhwc(g.contiguous())) for t, g in zip(self.taps, gouts) if g is not None)
        g, masked = None, False
        for si in range(top, -1, -1):
            blocks = self.stages[si]
            if g is None:
                g, masked = gmap.get(si), False
            g, masked = blocks_backward(K, blocks, g, masked, gmap.get(si - 1), self.stages[si - 1][-1] if si > 0 else None, True)
        K.end_backward(self.layers)
        return None if g is None else K.nhwc_to_nchw(g)


IMAGE_ARCHS = {   # name -> (block kind, blocks per layer, base widths, channel expansion)
    "resnet18": ("basic", (2, 2, 2, 2), (64, 128, 256, 512), 1),
    "resnet34": ("basic", (3, 4, 6, 3), (64, 128, 256, 512), 1),
    "resnet50": ("bottleneck", (3, 4, 6, 3), (64, 128, 256, 512), 4),
}


StackPlan._conv = None     # bound below to Plan._conv (same layer / parameter bookkeeping)


class Plan(object):
    """Builds the layer list, the parameter table and runs forward / backward."""

    def __init__(self, cfg, with_image=False, cf=64, image_arch="resnet18", image_blocks=None, image_widths=None):
        self.cfg = cfg
        self.with_image = with_image
        self.cf = cf
        self.table = ParamTable()
        self.layers = []
        lm = cfg["lidar_module"]
        self.widths = [lm["out_feature%d" % i] for i in range(1, 6)]
        self.nblocks = [lm["num_res_block%d" % i] for i in range(1, 6)]
        if cfg["voxel_channel"] != self.widths[0]:
            raise ValueError("voxel_channel must equal out_feature1 (model.py:67)")
        if cfg["voxel_length"] % 16 or cfg["voxel_width"] % 16:
            raise ValueError("voxel_length and voxel_width must be multiples of 16 (FPN add, model.py:151)")
        if any(w % 32 for w in self.widths) or cf % 32:
            raise ValueError("channel counts must be multiples of 32 (MFMA 32x32 tiles; dcf_weight_prep works on 32x32 weight tiles)")
        if len(set(self.widths)) != 5:
            raise ValueError("stage widths must differ (a stage strides only when its width changes, model.py:43-45)")
        self._build_lidar()
        self.fusion = []
        if with_image:
            if image_arch not in IMAGE_ARCHS:
                raise ValueError("fusion.image_stream must be one of %s (got %r)" % (sorted(IMAGE_ARCHS), image_arch))
            kind, nb, wd, exp = IMAGE_ARCHS[image_arch]
            self._build_image(image_blocks or nb, image_widths or wd, kind, exp)
            self._build_fusion()

    # ------------------------------------------------------------------ construction
    def _conv(self, name, cin, cout, k, stride, bn=None, kind="conv", need_dgrad=True, layout="ohwi", names=None, shapes=None):
        L = ConvLayer(len(self.layers), name, cin, cout, k[0], k[1], stride, k[0] // 2 if kind == "conv" else 0, bn, kind, need_dgrad, names)
        if kind == "stem":
            L.cin, L.kh, L.kw, L.pad = 32, 7, 1, 0
            L.w_off = self.table.add_param(name + ".weight", (cout, 3, 7, 7), "stem")
        elif names is not None:                 # several parameters laid out back to back (fused heads)
            offs = [self.table.add_param(n + ".weight", s, layout) for n, s in zip(names, shapes)]
            L.w_off = offs[0]
        elif kind == "linear":
            L.w_off = self.table.add_param(name + ".weight", (cout, cin), "plain")
        else:
            L.w_off = self.table.add_param(name + ".weight", (cout, cin, k[0], k[1]), layout)
        if bn is not None:
            L.gamma_off = self.table.add_param(bn + ".weight", (cout,))
            L.beta_off = self.table.add_param(bn + ".bias", (cout,))
            L.mean_off = self.table.add_buffer(bn + ".running_mean", (cout,))
            L.var_off = self.table.add_buffer(bn + ".running_var", (cout,))
            self.table.add_buffer(bn + ".num_batches_tracked", ())
        self.layers.append(L)
        return L

    def _build_lidar(self):
        """Key names and order of the reference's state_dict (model.py:64-79, :140-157)."""
        self.stages = []
        cin = self.widths[0]
        for si in range(5):
            cout = self.widths[si]
            blocks = []
            for bi in range(self.nblocks[si]):
                p = "lidar_backbone.backbone.layer%d.sequential.resblock_%d" % (si + 1, bi)
                ci = cin if bi == 0 else cout
                s = 2 if ci != cout else 1
                first = (si == 0 and bi == 0)
                c1 = self._conv(p + ".conv1", ci, cout, (3, 3), s, p + ".bn1", need_dgrad=not first)
                c2 = self._conv(p + ".conv2", cout, cout, (3, 3), 1, p + ".bn2")
                dn = self._conv(p + ".down_conv", ci, cout, (1, 1), 2, p + ".down_bn", need_dgrad=not first) if ci != cout else None
                blocks.append(Block(c1, c2, dn))
            for a_, b_ in zip(blocks[:-1], blocks[1:]):
                a_.next_conv = b_.conv1                # the consumer of a block's output inside its stage
            self.stages.append(blocks)
            cin = cout
        w = self.widths
        p = "lidar_backbone."
        self.latconv1 = self._conv(p + "latconv1", w[3], w[3], (1, 1), 1)
        self.downconv1 = self._conv(p + "downconv1", w[4], w[3], (1, 1), 1)
        self.latconv2 = self._conv(p + "latconv2", w[2], w[3], (1, 1), 1)
        self.conv3 = self._conv(p + "conv3", w[3], w[3], (3, 3), 1)
        # classconv (4) and bbox3dconv (14) share their input: one GEMM with 18 (->32) outputs
        self.heads = self._conv(p + "heads", w[3], 18, (1, 1), 1, names=[p + "classconv", p + "bbox3dconv"],
                                shapes=[(4, w[3], 1, 1), (14, w[3], 1, 1)])

    def _build_image(self, nblocks, widths, kind="basic", exp=1):
        """SURVEY.md App. D image stream: ResNet trunk (BasicBlock or Bottleneck, torchvision key names) + FPN."""
        p = "image_backbone"
        self.stem = self._conv(p + ".conv1", 3, widths[0], (7, 7), 2, p + ".bn1", kind="stem", need_dgrad=False)
        self.img_stages = []
        cin = widths[0]
        for li in range(4):
            w = widths[li]
            cout = w * exp
            blocks = []
            for bi in range(nblocks[li]):
                q = "%s.layer%d.%d" % (p, li + 1, bi)
                ci = cin if bi == 0 else cout
                s = 2 if (bi == 0 and li > 0) else 1
                down = bi == 0 and (li > 0 or ci != cout)
                dn = None
                if kind == "basic":
                    c1 = self._conv(q + ".conv1", ci, cout, (3, 3), s, q + ".bn1")
                    c2 = self._conv(q + ".conv2", cout, cout, (3, 3), 1, q + ".bn2")
                    if down:
                        dn = self._conv(q + ".downsample.0", ci, cout, (1, 1), s, q + ".downsample.1")
                    blocks.append(Block(c1, c2, dn))
                else:
                    c1 = self._conv(q + ".conv1", ci, w, (1, 1), 1, q + ".bn1")
                    c2 = self._conv(q + ".conv2", w, w, (3, 3), s, q + ".bn2")
                    c3 = self._conv(q + ".conv3", w, cout, (1, 1), 1, q + ".bn3")
                    if down:
                        dn = self._conv(q + ".downsample.0", ci, cout, (1, 1), s, q + ".downsample.1")
                    blocks.append(Bottleneck(c1, c2, c3, dn))
            for a_, b_ in zip(blocks[:-1], blocks[1:]):
                a_.next_conv = b_.conv1                # the consumer of a block's output inside its stage
            self.img_stages.append(blocks)
            cin = cout
        self.img_lat = [self._conv("image_fpn.lat%d" % (i + 1), widths[i] * exp, self.cf, (1, 1), 1) for i in range(4)]
        self.img_smooth = self._conv("image_fpn.smooth", self.cf, self.cf, (3, 3), 1)

    def _build_fusion(self):
        for si in range(1, 5):
            cb = self.widths[si]
            p = "fusion.site%d" % si
            f = {"stride": 2 ** si, "cb": cb}
            f["fc1_feat"] = self._conv(p + ".fc1_feat", self.cf, cb, (1, 1), 1, kind="linear")
            f["w1d_off"] = self.table.add_param(p + ".fc1_geo.weight", (cb, 3))
            f["b1_off"] = self.table.add_param(p + ".fc1.bias", (cb,))
            f["fc2"] = self._conv(p + ".fc2", cb, cb, (1, 1), 1, kind="linear")
            f["b2_off"] = self.table.add_param(p + ".fc2.bias", (cb,))
            self.fusion.append(f)

    # ------------------------------------------------------------------ forward
    def forward_image(self, K, x_image, save=True):
        """Camera stream alone (it does not depend on the frame's geometry): returns the feature map to hand to forward()
        as `fmap`.  The captured-graph path replays this part while the geometry still runs on its side stream."""
        self.ctx = {"save": save}
        fmap = self._image_forward(K, x_image, save)
        self._img_saved = self.ctx.get("img")
        return fmap

    def forward(self, K, x_lidar, x_image=None, geom=None, save=True, fmap=None, phase=None, resume=None):
        """x_lidar [B,Cz,L,W] fp32 NCHW (model.py:194); returns pred [B,32,L/4,W/4] fp32 NCHW.

        geom (fusion only): dict(xyz [B,n_max,3], uv [B,n_max,2], cnt [B] int32 device, idx = list over
        sites of [B,K,h,w] int32, aff) -- produced once per frame by the geometry kernels.
        fmap: the camera feature map when forward_image() already ran for this step.
        phase: None = the whole stream; 1 = only what does not need the KNN maps (input conversion, layer1, layer2's blocks:
        returns that activation); 2 = the rest, from `resume` = what phase 1 returned.  The captured-graph path replays
        phase 1 while the KNN still runs on the geometry side stream.
        """
        if phase != 2:
            self.ctx = {"save": save}
            if fmap is not None:
                if save:
                    self.ctx["img"] = self._img_saved
            elif self.with_image and geom is not None:
                fmap = self._image_forward(K, x_image, save)
                self.ctx["fmap_eager"] = fmap
            if geom is not None and geom.get("voxel_event") is not None:
                K.wait_event(geom["voxel_event"])          # voxel grid produced on the geometry side stream
            # a 16-bit x_lidar is already the input image [B,L,W,Cz] (train.geometry_async: written by the voxeliser)
            x = x_lidar if x_lidar.dtype != torch.float32 else K.nchw_to_nhwc(x_lidar)
            for si in (0, 1):
                x = blocks_forward(K, self.stages[si], x, save)
            if phase == 1:
                return x
        else:
            x = resume
            if fmap is None:
                fmap = self.ctx.pop("fmap_eager", None)
        self.ctx.pop("fmap_eager", None)
        outs = []
        for si in range(1, 5):
            if si > 1:
                x = blocks_forward(K, self.stages[si], x, save)
            if fmap is not None:
                x = self._fusion_forward(K, self.fusion[si - 1], x, fmap, geom, si - 1, save)
            outs.append(x)
        x1, x2, x3, x4 = outs
        l1 = K.conv_fwd(self.latconv1, x3, None, False)
        d1 = K.conv_fwd(self.downconv1, x4, None, False)
        t3 = K.resize_fwd(d1, (x3.shape[1], x3.shape[2]), True, l1)      # l1 + up(d1), model.py:161-163
        l2 = K.conv_fwd(self.latconv2, x2, None, False)
        t2 = K.resize_fwd(t3, (x2.shape[1], x2.shape[2]), True, l2)      # model.py:164-166
        xp = K.conv_fwd(self.conv3, t2, None, False)
        head = K.conv_fwd(self.heads, xp, None, False)
        pred = K.head_fwd(head, self.anchors_dev(K, head.shape[1], head.shape[2]))
        if save:
            self.ctx.update(x2=x2, x3=x3, x4=x4, t2=t2, xp=xp, head=head, pred=pred, d1_hw=(d1.shape[1], d1.shape[2]),
                            t3_hw=(t3.shape[1], t3.shape[2]), fused=fmap is not None)
        return pred

    def anchors_dev(self, K, h, w):
        key = (h, w)
        if getattr(self, "_anc_key", None) != key:
            from .model import AnchorBoundingBoxFeature
            anc = AnchorBoundingBoxFeature(self.cfg)()
            if tuple(anc.shape[-2:]) != (h, w):
                raise ValueError("anchor grid %s does not match the head output %s (reduced_scale must be 4)" % (tuple(anc.shape[-2:]), key))
            self._anc = K.to_device(anc.contiguous())
            self._anc_key = key
        return self._anc

    # ------------------------------------------------------------------ backward
    def backward(self, K, gpred):
        """gpred [B,32,h,w] fp32.  Fills the gradient arena (through the backend)."""
        c = self.ctx
        K.begin_backward(self.layers)
        ghead = K.head_bwd(c["head"], self.anchors_dev(K, c["head"].shape[1], c["head"].shape[2]), c["pred"], gpred)
        K.conv_wgrad(self.heads, c["xp"], ghead)
        gxp = K.conv_dgrad(self.heads, ghead, tuple(c["xp"].shape), None)
        K.conv_wgrad(self.conv3, c["t2"], gxp)
        gt2 = K.conv_dgrad(self.conv3, gxp, tuple(c["t2"].shape), None)
        K.conv_wgrad(self.latconv2, c["x2"], gt2)
        gx2 = K.conv_dgrad(self.latconv2, gt2, tuple(c["x2"].shape), None)
        gt3 = K.resize_bwd(gt2, c["t3_hw"], True)
        K.conv_wgrad(self.latconv1, c["x3"], gt3)
        gx3 = K.conv_dgrad(self.latconv1, gt3, tuple(c["x3"].shape), None)
        gd1 = K.resize_bwd(gt3, c["d1_hw"], True)
        K.conv_wgrad(self.downconv1, c["x4"], gd1)
        g = K.conv_dgrad(self.downconv1, gd1, tuple(c["x4"].shape), None)
        extras = {3: gx3, 2: gx2}          # gradient joining the output of stage index 3 / 2
        gF = None
        masked = False
        for si in range(4, -1, -1):
            if si >= 1 and c["fused"]:
                last = self.stages[si][-1]
                if FUSED_SITE_MASK and type(last) is Block and last.saved is not None and hasattr(K, "relu_mask_rowscale_bwd"):
                    # one pass over the site's gradient: fc2's bias gradient (cnt-weighted channel sums of the UNMASKED g) and the
                    # masked copy the stage's last block goes on with (round 6: was rowscale_bias_bwd here + relu_mask in place there)
                    f = self.fusion[si - 1]
                    gm = K.relu_mask_rowscale_bwd(g, last.saved[2], c["fuse%d" % (si - 1)]["cnt"], f["b2_off"])
                    gF = self._fusion_backward(K, f, g, si - 1, gF, bias_done=True)
                    g, masked = gm, True
                else:
                    gF = self._fusion_backward(K, self.fusion[si - 1], g, si - 1, gF)
            # the block feeding this stage's first one: the previous stage's last block when no fusion site sits in between
            # (the fusion backward needs the unmasked gradient)
            prev0 = self.stages[si - 1][-1] if (si > 0 and not (c["fused"] and si - 1 >= 1)) else None
            g, masked = blocks_backward(K, self.stages[si], g, masked, extras.get(si - 1), prev0, need_gx0=(si != 0))
            if si == 3:
                # stages 4-5, the FPN and the heads are complete: two thirds of the LiDAR stream's parameters (data-parallel
                # runs start that bucket's all-reduce here)
                K.bucket_ready(self.layers, "lidar_hi")
        if c["fused"]:
            # every gradient of the LiDAR stream and of the fusion layers is complete here; only the camera stream is left:
            # the backend may finalise and hand over that bucket now (data-parallel runs all-reduce it under the camera
            # stream's backward)
            K.bucket_ready(self.layers, "lidar+fusion")
            self._image_backward(K, gF)
        K.end_backward(self.layers)
        self.ctx = {}

    # ------------------------------------------------------------------ image stream
    def _image_forward(self, K, x_image, save):
        B, _, Hh, W = x_image.shape
        img4 = K.image_to_nhwc4(x_image)
        c1 = K.stem_fwd(self.stem, img4, Hh, W)
        if save:
            x, pool_idx = K.maxpool_fwd_idx(c1)
        else:
            x, pool_idx = K.maxpool_fwd(c1), None
        feats = []
        for blocks in self.img_stages:
            x = blocks_forward(K, blocks, x, save)
            feats.append(x)
        c2, c3, c4, c5 = feats
        p = K.conv_fwd(self.img_lat[3], c5, None, False)
        for i in (2, 1, 0):
            lat = K.conv_fwd(self.img_lat[i], feats[i], None, False)
            p = K.resize_fwd(p, (feats[i].shape[1], feats[i].shape[2]), False, lat)
            if i == 0:
                p2 = p
        fmap = K.conv_fwd(self.img_smooth, p2, None, False)
        if save:
            self.ctx["img"] = dict(img4=img4, hw=(Hh, W), c1=c1, pool_idx=pool_idx, feats=feats, p2=p2)
        return fmap

    def _image_backward(self, K, gF):
        im = self.ctx["img"]
        feats = im["feats"]
        g = K.cast_like(gF, im["p2"])                      # fp32 accumulator -> compute dtype
        K.conv_wgrad(self.img_smooth, im["p2"], g)
        gp = K.conv_dgrad(self.img_smooth, g, tuple(im["p2"].shape), None)
        gfeat = [None] * 4
        for i in (0, 1, 2):
            K.conv_wgrad(self.img_lat[i], feats[i], gp)
            gfeat[i] = K.conv_dgrad(self.img_lat[i], gp, tuple(feats[i].shape), None)
            gp = K.resize_bwd(gp, (feats[i + 1].shape[1], feats[i + 1].shape[2]), False)
        K.conv_wgrad(self.img_lat[3], feats[3], gp)
        g = K.conv_dgrad(self.img_lat[3], gp, tuple(feats[3].shape), None)
        masked = False
        for li in range(3, -1, -1):
            g, masked = blocks_backward(K, self.img_stages[li], g, masked, gfeat[li - 1] if li > 0 else None,
                                        self.img_stages[li - 1][-1] if li > 0 else None, True)
            if li == 3:
                K.bucket_ready(self.layers, "image_hi")        # camera layer4 + FPN: three quarters of the camera stream's parameters
        # g = gradient at the max-pool output
        gc1 = K.maxpool_bwd(im["c1"], g, None, im.get("pool_idx"))
        gc1 = K.bn_bwd(self.stem, K.relu_mask(gc1, im["c1"]))
        K.stem_wgrad(self.stem, im["img4"], gc1, im["hw"][0], im["hw"][1])

    # ------------------------------------------------------------------ fusion
    def _fusion_forward(self, K, f, x, fmap, geom, site, save):
        B, h, w, cb = x.shape
        if geom.get("event") is not None and not geom.get("_waited"):
            K.wait_event(geom["event"])                # KNN indices produced on the geometry side stream
            geom["_waited"] = True
        n_max = self._fusion_rows(geom)
        # every site samples the same camera map at the same (u, v): the point features are computed once per step
        fp = self.ctx.get("fuse_fp") if save else None
        if fp is None:
            fp = K.point_sample_fwd(fmap, geom["uv"], geom["cnt"], n_max)        # [B,n_max,Cf]
            if save:
                self.ctx["fuse_fp"] = fp
        P = K.conv_fwd(f["fc1_feat"], fp.view(B, n_max, 1, fp.shape[-1]), None, False).view(B, n_max, cb)
        hsum, cnt = K.fusion_gather_fwd(P, geom["xyz"], geom["idx"][site], f["stride"], geom["aff"], f["w1d_off"], f["b1_off"])
        out = K.conv_fwd_rowscale(f["fc2"], hsum, x, cnt, f["b2_off"])           # x + hsum.W2^T + cnt*b2 (one epilogue)
        if save:
            self.ctx["fuse%d" % site] = dict(fp=fp, P=P, hsum=hsum, cnt=cnt, geom=geom, fmap_shape=tuple(fmap.shape))
        return out

    @staticmethod
    def _fusion_rows(geom):
        """Rows of the per-point fusion tensors.  xyz / uv are padded to max_num_pc, but only the first n_valid rows of a
        frame are real (typically a third: the points inside the camera frustum).  When the producer of the geometry
        handed over a host copy of the counts (train.geometry_async: pinned buffer + event, issued before the KNN), the
        per-point tensors -- sampled features, the fc1 GEMMs, their gradients -- are sized to the largest count rounded
        up to 256 rows instead.  The wait is on a copy that finished long before the host gets here (the host runs a
        couple of ms ahead of the GPU)."""
        n = geom.get("n_rows")
        if n is None:
            n = geom["xyz"].shape[1]
            ch = geom.get("cnt_host")
            if ch is not None:
                geom["cnt_event"].synchronize()
                n = min(n, max(256, (int(ch.max()) + 255) // 256 * 256))
            geom["n_rows"] = n
        return n

    def _fusion_backward(self, K, f, g, site, gF, bias_done=False):
        s = self.ctx.pop("fuse%d" % site)
        geom = s["geom"]
        B, n_max, cb = s["P"].shape
        if not bias_done:
            K.rowscale_bias_bwd(g, s["cnt"], f["b2_off"])
        # (two-pass form: g is masked in place by the stage's last block afterwards, so fc2's weight gradient cannot wait for the
        # grouped launch; one-pass form: g stays as it is and the layer joins the group)
        K.conv_wgrad(f["fc2"], s["hsum"], g, defer=bias_done)
        ghsum = K.conv_dgrad(f["fc2"], g, tuple(s["hsum"].shape), None)
        inv = geom.get("inv") if FUSION_INV != "2" else None
        if inv and geom.get("inv_event") is not None and not geom.get("_inv_waited"):
            K.wait_event(geom["inv_event"])            # inverse KNN maps produced on the geometry side stream
            geom["_inv_waited"] = True
        gP = K.fusion_gather_bwd(s["P"], geom["xyz"], geom["idx"][site], f["stride"], geom["aff"], f["w1d_off"], f["b1_off"], ghsum,
                                 inv, site, geom.get("inv_nmax"))
        gPc = K.cast_like(gP, s["P"]).view(B, n_max, 1, cb)
        fp4 = s["fp"].view(B, n_max, 1, s["fp"].shape[-1])
        K.conv_wgrad(f["fc1_feat"], fp4, gPc)
        # dL/dfp is summed over the sites through the dgrad kernel's residual input and scattered into the camera map
        # once, after the last site (site 0) has added its share
        gfp = K.conv_dgrad(f["fc1_feat"], gPc, tuple(fp4.shape), self.ctx.get("fuse_gfp"))
        if site > 0:
            self.ctx["fuse_gfp"] = gfp
            return None
        self.ctx.pop("fuse_gfp", None)
        return K.point_sample_bwd(gfp.view(B, n_max, -1), geom["uv"], geom["cnt"], n_max, s["fmap_shape"], gF)


StackPlan._conv = Plan._conv
