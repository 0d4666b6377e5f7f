"""CPU suite: the engine's chain SEQUENCING (engine.blocks_forward / blocks_backward) against its per-block path, with a tiny
torch-CPU stand-in for the backend (test infrastructure: the product's backend is HipBackend and nothing else).  What is checked is
the bookkeeping a chain launch gets -- which layer reads which, which residual / mask each layer takes (a tensor or the index of
an earlier chain layer), which (x, gy) pairs are queued for the weight gradients, what each block saves -- for stages with and
without a strided first block (/root/reference/model.py:32-45, :48-60)."""
import importlib

import torch
import torch.nn.functional as F

from _util import PKG

E = importlib.import_module(PKG + ".engine")


class FakeBackend(object):
    """NHWC fp32 tensors; weights per layer name; records the weight-gradient queue."""

    def __init__(self, layers, chain):
        g = torch.Generator().manual_seed(3)
        self.w = {L.name: (torch.rand((L.cout, L.cin, L.kh, L.kw), generator=g) - 0.5) * 0.2 for L in layers}
        self.chain = chain
        self.wq = []
        self.chain_calls = 0

    # ---- what the engine calls
    def can_chain(self, shape, n):
        return self.chain and n >= 2

    def conv_fwd(self, L, x, res, relu, nxt=None):
        y = F.conv2d(x.permute(0, 3, 1, 2), self.w[L.name], None, L.stride, L.pad).permute(0, 2, 3, 1)
        if res is not None:
            y = y + res
        L.out_shape = tuple(y.shape[:3])
        return torch.relu(y) if relu else y

    def conv_dgrad(self, L, gy, in_shape, res, mask=None):
        x = torch.zeros(in_shape, requires_grad=True)
        y = F.conv2d(x.permute(0, 3, 1, 2), self.w[L.name], None, L.stride, L.pad).permute(0, 2, 3, 1)
        (gx,) = torch.autograd.grad(y, x, gy)
        if res is not None:
            gx = gx + res
        return gx * (mask > 0) if mask is not None else gx

    def shortcut_dgrad(self, Ld, dd, L1, g1, in_shape, mask=None):
        return self.conv_dgrad(L1, g1, in_shape, self.conv_dgrad(Ld, dd, in_shape, None), mask)

    def chain_fwd(self, x, specs):
        self.chain_calls += 1
        outs, cur = [], x
        for (L, r, relu) in specs:
            assert not isinstance(r, int) or r < len(outs) - 0, "a residual must come from an earlier layer"
            cur = self.conv_fwd(L, cur, outs[r] if isinstance(r, int) else r, relu)
            outs.append(cur)
        return outs

    def chain_dgrad(self, g, specs):
        self.chain_calls += 1
        outs, cur = [], g
        for (L, r, m) in specs:
            cur = self.conv_dgrad(L, cur, tuple(cur.shape), outs[r] if isinstance(r, int) else r, outs[m] if isinstance(m, int) else m)
            outs.append(cur)
        return outs

    def conv_wgrad(self, L, x, gy, defer=True):
        self.wq.append((L.name, x.clone(), gy.clone()))

    def bn_bwd(self, L, g):
        return g

    def relu_mask(self, g, y):
        g.mul_((y > 0).to(g.dtype))
        return g


def _stage(strided, n, c_in=8, c=16):
    """n blocks; the first one strided (conv1 stride 2 + 1x1 shortcut) or an identity block like the rest."""
    layers, blocks = [], []
    for b in range(n):
        ci = c_in if (b == 0 and strided) else c
        s = 2 if (b == 0 and strided) else 1
        c1 = E.ConvLayer(len(layers), "b%d.conv1" % b, ci, c, 3, 3, s, 1); layers.append(c1)
        c2 = E.ConvLayer(len(layers), "b%d.conv2" % b, c, c, 3, 3, 1, 1); layers.append(c2)
        dn = None
        if b == 0 and strided:
            dn = E.ConvLayer(len(layers), "b%d.down" % b, ci, c, 1, 1, 2, 0); layers.append(dn)
        for L in (c1, c2, dn):
            if L is not None:
                L.cout_pad = L.cout
        blocks.append(E.Block(c1, c2, dn))
    return layers, blocks


def _run(strided, n, chain, prev0, extra):
    layers, blocks = _stage(strided, n)
    K = FakeBackend(layers, chain)
    g = torch.Generator().manual_seed(11)
    x = torch.rand((2, 12, 10, 8 if strided else 16), generator=g) - 0.3
    y = E.blocks_forward(K, blocks, x.clone(), save=True)
    saved = [tuple(t.clone() for t in b.saved) for b in blocks]
    gy = torch.rand(tuple(y.shape), generator=g) - 0.5
    ex = (torch.rand(tuple(x.shape), generator=g) - 0.5) if extra else None
    p0 = E.Block(None, None) if prev0 else None              # (only its identity is used: "a block feeds this stage")
    gx, masked = E.blocks_backward(K, blocks, gy.clone(), False, ex, p0, True)
    return y, saved, gx, masked, K


def _same(a, b):
    return a.shape == b.shape and float((a - b).abs().max()) < 1e-5


def test_chain_sequencing_equals_per_block_path():
    for strided in (True, False):
        for n in (1, 2, 4):
            for prev0 in (False, True):
                for extra in ((False, True) if strided else (False,)):        # an identity-shortcut block takes no extra gradient
                    y0, s0, gx0, m0, K0 = _run(strided, n, False, prev0, extra)
                    y1, s1, gx1, m1, K1 = _run(strided, n, True, prev0, extra)
                    tag = (strided, n, prev0, extra)
                    nchain = 2 * n - 1 if strided else 2 * n
                    assert K0.chain_calls == 0 and K1.chain_calls == (2 if nchain >= 2 else 0), tag
                    assert _same(y0, y1), tag
                    assert all(_same(a, b) for sa, sb in zip(s0, s1) for a, b in zip(sa, sb)), tag      # what each block saved
                    assert m0 == m1 and _same(gx0, gx1), tag
                    q0 = sorted((nm, x, g) for nm, x, g in K0.wq)
                    q1 = sorted((nm, x, g) for nm, x, g in K1.wq)
                    assert [t[0] for t in q0] == [t[0] for t in q1], tag                                 # every layer queued once
                    assert all(_same(a[1], b[1]) and _same(a[2], b[2]) for a, b in zip(q0, q1)), tag
