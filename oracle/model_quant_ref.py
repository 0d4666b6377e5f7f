"""TEST INFRASTRUCTURE -- quantisation-aware CPU statement of the LiDAR stream (eval-mode BatchNorm).

Only tests/ may import this module; the product package never does.

oracle/model_ref.py restates /root/reference/model.py:10-204 in fp32.  The HIP path's 16-bit modes (bf16 = the
benchmarked type, fp16) keep fp32 accumulators and fp32 master weights but STORE activations, folded weights and
activation gradients in the 16-bit type.  Comparing such a run with the fp32 statement needs tolerances of several
per cent, which could hide a wrong tap or a dropped tile.  This file states the same network with a rounding to the
storage type at exactly the places where the device rounds, forward AND backward, so that the comparison isolates
implementation errors from quantisation noise (what is left: fp32 summation order, i.e. rare one-ulp flips):

  forward   x -> Q; folded weight Q(scale*W), shift fp32; every conv epilogue Q(act(acc + shift + residual));
            FPN Q(lateral + upsample(.)); head tensor Q; softmax / box decode in fp32 (model.py:159-173,126-136)
  backward  every activation gradient is rounded once where the device stores it: the head gradient, each dgrad
            output after its fused residual-gradient add (+ ReLU mask, which commutes with the rounding), each
            resize backward; weight gradients are fp32 sums of products of the rounded operands; the folded-BN chain
            rule (dW = scale*G, dbeta = sum g, dgamma = (<W,G> - mean*dbeta)*rsqrt(var+eps)) falls out of autograd.
  The one place where the device rounds twice is the input gradient of a stage's first block when an FPN lateral
  joins there (engine.Block.backward: Q(dgrad_down + extra), then Q(dgrad_conv1 + that)); `_qgrad` reproduces it.

Same pinning as model_ref (it IS model_ref with rounding points; with qdtype=None the two agree to fp32 noise, which
tests/test_oracle_golden.py checks against the reference's golden vectors).
"""
import torch
import torch.nn.functional as F

from . import model_ref

BN_EPS = 1e-5


class _QPoint(torch.autograd.Function):
    """Storage point of an activation: value rounded to the storage type; its gradient (summed over all consumers in
    fp32 by autograd, like the device's fused epilogue adds) is rounded to the storage type as well."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(torch.float32), None


class _QGrad(torch.autograd.Function):
    """Identity whose gradient is rounded: an extra rounding point of the backward pass only."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(torch.float32), None


class _QSte(torch.autograd.Function):
    """Weight image: rounded forward, gradient passed through to the fp32 master weight."""

    @staticmethod
    def forward(ctx, w, dt):
        return w.to(dt).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g, None


def _q(x, dt):
    return x if dt is None else _QPoint.apply(x, dt)


def _qgrad(x, dt):
    return x if dt is None else _QGrad.apply(x, dt)


def _conv_bn(sd, conv, bn, x, stride, pad, dt):
    """Convolution with the eval-mode BatchNorm folded in the way dcf_weight_prep does: Q(scale*W), fp32 shift."""
    w = sd[conv + ".weight"]
    if bn is None:
        wq = w if dt is None else _QSte.apply(w, dt)
        return F.conv2d(x, wq, None, stride, pad)
    scale = sd[bn + ".weight"] * torch.rsqrt(sd[bn + ".running_var"] + BN_EPS)
    shift = sd[bn + ".bias"] - sd[bn + ".running_mean"] * scale
    ws = w * scale.view(-1, 1, 1, 1)
    wq = ws if dt is None else _QSte.apply(ws, dt)
    return F.conv2d(x, wq, None, stride, pad) + shift.view(1, -1, 1, 1)


def _resblock(sd, pfx, x, dt):
    """model.py:32-41 with the device's storage points.  Returns (y, x as the shortcut branch sees it): where an FPN
    lateral also reads x, its gradient and the shortcut convolution's are summed and rounded first, then the first
    convolution's share is added and rounded again (engine.Block.backward)."""
    w1 = sd[pfx + ".conv1.weight"]
    change = w1.shape[0] != w1.shape[1]
    s = 2 if change else 1
    if change:
        xs = _qgrad(x, dt)                      # Q(dgrad_down + extra): engine.Block.backward, `gx`
        r = _q(_conv_bn(sd, pfx + ".down_conv", pfx + ".down_bn", xs, 2, 0, dt), dt)
    else:
        xs = x
        r = x
    y1 = _q(F.relu(_conv_bn(sd, pfx + ".conv1", pfx + ".bn1", x, s, 1, dt)), dt)
    y = _q(F.relu(_conv_bn(sd, pfx + ".conv2", pfx + ".bn2", y1, 1, 1, dt) + r), dt)
    return y, xs


def forward(sd, cfg, x_lidar, qdtype=None):
    """LiDAR-only forward (model.py:194-204), eval-mode BN.  qdtype: torch.bfloat16 / torch.float16 / None (= fp32,
    no rounding).  sd values that require grad receive the gradients the device's backward would produce."""
    dt = qdtype
    pre = "lidar_backbone."
    bb = pre + "backbone."
    x = _q(x_lidar, dt)
    outs = []
    lat_src = {}
    for si, name in enumerate(model_ref.STAGES):
        i = 0
        while (bb + name + ".sequential.resblock_%d.conv1.weight" % i) in sd:
            x, xs = _resblock(sd, bb + name + ".sequential.resblock_%d" % i, x, dt)
            if i == 0 and si >= 1:
                lat_src[si - 1] = xs            # the previous stage's output as the shortcut branch sees it
            i += 1
        if si >= 1:
            outs.append(x)
    # FPN laterals read a stage output through the same gradient-rounding point as the next stage's shortcut conv
    # (x2 = layer3 output feeds layer4; x3 = layer4 output feeds layer5; x4 = layer5 output feeds only the FPN)
    x2 = lat_src[2]
    x3 = lat_src[3]
    x4 = outs[3]
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)      # model.py:149
    l1 = _q(_conv_bn(sd, pre + "latconv1", None, _q(x3, dt), 1, 0, dt), dt)
    d1 = _q(_conv_bn(sd, pre + "downconv1", None, x4, 1, 0, dt), dt)
    t3 = _q(l1 + up(d1), dt)
    l2 = _q(_conv_bn(sd, pre + "latconv2", None, _q(x2, dt), 1, 0, dt), dt)
    t2 = _q(l2 + up(t3), dt)
    xp = _q(_conv_bn(sd, pre + "conv3", None, t2, 1, 1, dt), dt)
    # the two heads are one GEMM on the device; its output tensor is stored in the compute type
    cls = _q(_conv_bn(sd, pre + "classconv", None, xp, 1, 0, dt), dt)
    reg = _q(_conv_bn(sd, pre + "bbox3dconv", None, xp, 1, 0, dt), dt)
    cls = torch.cat((F.softmax(cls[:, 0:2], 1), F.softmax(cls[:, 2:4], 1)), 1)
    box = model_ref.decode(reg, model_ref.anchors(cfg))
    return torch.cat((cls, reg, box), 1)
