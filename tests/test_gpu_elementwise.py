"""GPU parity: the HBM-bound kernels around the convolutions, against torch fp32 CPU ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import TORCH_DT, from_dev, golden_cfg, load_golden, pkg, q, rel_err, rnd, to_dev
from oracle import model_ref

pytestmark = pytest.mark.gpu
TOL = {0: 1e-5, 1: 1e-2, 2: 2e-3}


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_nchw_to_nhwc(dtype):
    ops = pkg("ops")
    x = rnd((2, 32, 9, 7), 1)
    y = ops.nchw_to_nhwc(x.cuda(), dtype)
    assert torch.equal(from_dev(y), q(x, dtype))
    x = rnd((1, 12, 5, 6), 2)
    assert torch.equal(from_dev(ops.nchw_to_nhwc(x.cuda(), dtype)), q(x, dtype))


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_relu_bwd_chansum(dtype):
    ops = pkg("ops")
    for C in (32, 96, 192):
        y = q(torch.relu(rnd((2, C, 11, 13), 3)), dtype)
        gy = q(rnd((2, C, 11, 13), 4), dtype)
        gsum = torch.zeros(C, device="cuda")
        g = to_dev(gy, dtype)
        ops.relu_bwd_chansum(dtype, g, to_dev(y, dtype), gsum, True)
        ref = gy * (y > 0)
        assert torch.equal(from_dev(g), ref)
        assert torch.allclose(gsum.cpu(), ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
        gsum.zero_()
        g2 = to_dev(gy, dtype)
        ops.relu_bwd_chansum(dtype, g2, None, gsum, False)
        assert torch.allclose(gsum.cpu(), gy.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("C,hw", [(64, (300, 400)), (36, (150, 130)), (256, (90, 100))])
def test_relu_bwd_chansum_grid_stride_trips(C, hw, dtype):
    """Sizes at which a thread walks several elements (two per trip + a tail) and the 4-channel instantiation (C = 36)."""
    ops = pkg("ops")
    y = q(torch.relu(rnd((2, C) + hw, 31)), dtype)
    gy = q(rnd((2, C) + hw, 32), dtype)
    gsum = torch.zeros(C, device="cuda")
    g = to_dev(gy, dtype)
    ops.relu_bwd_chansum(dtype, g, to_dev(y, dtype), gsum, True)
    ref = gy * (y > 0)
    assert torch.equal(from_dev(g), ref)
    want = ref.double().sum((0, 2, 3))
    assert float((gsum.cpu().double() - want).abs().max()) <= 1e-4 * float(ref.abs().double().sum((0, 2, 3)).max())


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("C,npix", [(64, 70000), (36, 9001), (256, 17600), (128, 300)])
def test_rowscale_bias_fwd_bwd(C, npix, dtype):
    """fc2's bias under the neighbour sum (model.py:216-219: sum_k (W2 h_k + b2) = W2 sum_k h_k + cnt * b2): y += cnt[p] * b2[c]
    and db2[c] += sum_p cnt[p] * g[p, c], against fp64 sums."""
    ops = pkg("ops")
    g = torch.Generator().manual_seed(5)
    cnt = torch.randint(0, 4, (npix,), generator=g).float()
    b2 = rnd((C,), 41)
    y = q(rnd((npix, C), 42), dtype)
    yd = to_dev(y.view(1, npix, 1, C).permute(0, 3, 1, 2), dtype).view(npix, C)
    out = ops.rowscale_bias_fwd(dtype, yd, cnt.cuda(), b2.cuda())
    want = y + cnt[:, None] * b2[None, :]
    assert rel_err(out.float().cpu(), want) < TOL[dtype]
    gy = q(rnd((npix, C), 43), dtype)
    gd = to_dev(gy.view(1, npix, 1, C).permute(0, 3, 1, 2), dtype).view(npix, C)
    gb = torch.full((C,), 0.5, device="cuda")
    ops.rowscale_bias_bwd(dtype, gd, cnt.cuda(), gb)
    ref = (cnt[:, None].double() * gy.double()).sum(0) + 0.5
    bound = 1e-5 * float((cnt[:, None].double() * gy.double().abs()).sum(0).max()) + 1e-6
    assert float((gb.cpu().double() - ref).abs().max()) <= bound


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("C,npix", [(64, 70000), (36, 9001), (256, 17600), (128, 300), (192, 2 * 88 * 100)])
def test_relu_mask_rowscale_bwd(C, npix, dtype):
    """A fusion site's one-pass form (round 6): gout = g * (y > 0) in a tensor of its own, bit for bit what the in-place mask writes,
    g itself untouched; db2[c] += sum_p cnt[p] * g[p, c] of the UNMASKED g, against fp64 sums (the two-pass form's bound)."""
    ops = pkg("ops")
    gen = torch.Generator().manual_seed(6)
    cnt = torch.randint(0, 4, (npix,), generator=gen).float()
    gy = q(rnd((npix, C), 44), dtype)
    y = q(torch.relu(rnd((npix, C), 45)), dtype)                 # about half of the activations are exactly zero
    gd = to_dev(gy.view(1, npix, 1, C).permute(0, 3, 1, 2), dtype).view(npix, C)
    yd = to_dev(y.view(1, npix, 1, C).permute(0, 3, 1, 2), dtype).view(npix, C)
    keep = gd.clone()
    gb = torch.full((C,), 0.25, device="cuda")
    gout = ops.relu_mask_rowscale_bwd(dtype, gd, yd, cnt.cuda(), gb)
    assert gout.data_ptr() != gd.data_ptr() and torch.equal(gd, keep)
    two_pass = keep.clone()
    ops.relu_bwd_chansum(dtype, two_pass, yd, None, True)       # the in-place mask of the two-pass form
    assert torch.equal(gout, two_pass)
    assert torch.equal(gout.float().cpu(), torch.where(y > 0, gy, torch.zeros_like(gy)))
    ref = (cnt[:, None].double() * gy.double()).sum(0) + 0.25
    bound = 1e-5 * float((cnt[:, None].double() * gy.double().abs()).sum(0).max()) + 1e-6
    assert float((gb.cpu().double() - ref).abs().max()) <= bound


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("case", [((6, 4), (12, 8), True), ((12, 39), (24, 78), False), ((24, 78), (47, 156), False), ((5, 7), (5, 7), False),
                                  ((44, 50), (88, 100), True), ((5, 6), (23, 31), True), ((4, 5), (19, 26), False),      # x4-5: more matches than the register list holds
                                  ((20, 30), (9, 13), False), ((21, 17), (8, 6), True)])                                # downsampling
def test_resize_bilinear(case, dtype):
    ops = pkg("ops")
    (hi, wi), (ho, wo), ac = case
    x = q(rnd((2, 64, hi, wi), 7), dtype).requires_grad_(True)
    add = q(rnd((2, 64, ho, wo), 8), dtype)
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=ac)
    y = ops.resize_bilinear_fwd(dtype, to_dev(x.detach(), dtype), (ho, wo), ac, to_dev(add, dtype))
    assert rel_err(from_dev(y), (ref + add).detach()) < TOL[dtype]
    gy = q(rnd((2, 64, ho, wo), 9), dtype)
    ref.backward(gy)
    gx = ops.resize_bilinear_bwd(dtype, to_dev(gy, dtype), (hi, wi), ac)
    assert rel_err(from_dev(gx), x.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1])
def test_resize_bilinear_bwd_four_channel_granularity(dtype):
    """Channel counts that are multiples of 4 but not of 8 take the 4-wide instantiation."""
    ops = pkg("ops")
    x = q(rnd((2, 36, 9, 11), 17), dtype).requires_grad_(True)
    ref = F.interpolate(x, size=(18, 22), mode="bilinear", align_corners=True)
    gy = q(rnd((2, 36, 18, 22), 19), dtype)
    ref.backward(gy)
    gx = ops.resize_bilinear_bwd(dtype, to_dev(gy, dtype), (9, 11), True)
    assert rel_err(from_dev(gx), x.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_maxpool(dtype):
    ops = pkg("ops")
    for (hh, ww) in ((12, 16), (19, 31)):
        x = q(rnd((2, 64, hh, ww), 10), dtype).requires_grad_(True)
        ref = F.max_pool2d(x, 3, 2, 1)
        xd = to_dev(x.detach(), dtype)
        y = ops.maxpool_fwd(dtype, xd)
        assert torch.equal(from_dev(y), ref.detach())
        gy = q(rnd(tuple(ref.shape), 11), dtype)
        ref.backward(gy)
        gx = ops.maxpool_bwd(dtype, xd, y, to_dev(gy, dtype))            # with the pooled output: equality test + tie scan
        assert rel_err(from_dev(gx), x.grad) < TOL[dtype]
        gx0 = ops.maxpool_bwd(dtype, xd, None, to_dev(gy, dtype))        # without: arg-max recomputed per window
        assert torch.equal(gx0, gx)
        y2, idx = ops.maxpool_fwd_idx(dtype, xd)                         # arg-max recorded by the forward: plain gather
        assert torch.equal(y2, y) and int(idx.view(torch.uint8).max()) <= 8
        assert torch.equal(ops.maxpool_bwd_idx(dtype, idx, to_dev(gy, dtype), tuple(xd.shape)), gx)


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_head_fwd_bwd(dtype):
    """softmax pairs + box decode + concat against the restated model.py:116-137,168-172,204."""
    ops = pkg("ops")
    cfg = golden_cfg(load_golden("model_tiny.npz"))
    anc = model_ref.anchors(cfg)
    B, h, w = 2, 16, 8
    head = q(rnd((B, 32, h, w), 12, -1.5, 1.5), dtype)
    hv = head[:, :18].clone().requires_grad_(True)
    cls = torch.cat((torch.softmax(hv[:, 0:2], 1), torch.softmax(hv[:, 2:4], 1)), 1)
    reg = hv[:, 4:18]
    ref = torch.cat((cls, reg, model_ref.decode(reg, anc)), 1)
    pred = ops.head_fwd(dtype, to_dev(head, dtype), anc.cuda())
    assert rel_err(pred.cpu(), ref.detach()) < 1e-5
    R = rnd((B, 32, h, w), 13)
    ref.backward(R)
    gh = ops.head_bwd(dtype, to_dev(head, dtype), anc.cuda(), pred, R.cuda())
    got = from_dev(gh)
    assert rel_err(got[:, :18], hv.grad) < TOL[dtype]
    assert float(got[:, 18:].abs().max()) == 0.0


def test_head_matches_golden_decode():
    ops = pkg("ops")
    z = load_golden("anchors_decode.npz")
    reg = torch.from_numpy(z["reg"])
    head = torch.zeros(2, 32, 16, 8)
    head[:, 4:18] = reg
    pred = ops.head_fwd(0, to_dev(head, 0), torch.from_numpy(z["anchors_tiny"]).cuda()).cpu().numpy()
    assert np.abs(pred[:, 18:32] - z["box"]).max() <= 1e-5 * np.abs(z["box"]).max()
    assert np.array_equal(pred[:, 4:18], z["reg"])
    assert np.allclose(pred[:, 0:4], 0.5)


def test_adam_matches_torch():
    ops = pkg("ops")
    n = 4099
    p0, g = rnd((n,), 14), rnd((n,), 15)
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.999))
    pd, m, v = p0.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gg = g * step
        p.grad = gg.clone()
        opt.step()
        ops.adam_step(pd, gg.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, step)
        assert torch.allclose(pd.cpu(), p.detach(), rtol=1e-6, atol=1e-7), step


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_weight_prep_and_finalize(dtype):
    """Table-driven BN folding and the folded-BN gradient chain rule against autograd."""
    H, ops = pkg("_hip"), pkg("ops")
    import ctypes
    Cout, Cin, k, cp = 18, 32, 1, 32       # a padded head-like conv without BN
    C2o, C2i, k2 = 64, 32, 3               # a conv with BN
    w1, w2 = rnd((Cout, k, k, Cin), 16, -0.3, 0.3), rnd((C2o, k2, k2, C2i), 17, -0.3, 0.3)
    gamma, beta = rnd((C2o,), 18, 0.5, 1.5), rnd((C2o,), 19, -0.2, 0.2)
    mean, var = rnd((C2o,), 20, -0.2, 0.2), rnd((C2o,), 21, 0.5, 1.5)
    params = torch.cat((w1.reshape(-1), w2.reshape(-1), gamma, beta)).cuda()
    buffers = torch.cat((mean, var)).cuda()
    o_w1, o_w2 = 0, w1.numel()
    o_g, o_b = o_w2 + w2.numel(), o_w2 + w2.numel() + C2o
    es = 4 if dtype == 0 else 2
    wf1, wd1 = 0, cp * Cin * es
    wf2 = wd1 + cp * Cin * es
    wd2 = wf2 + w2.numel() * es
    wbytes = wd2 + w2.numel() * es
    ns1, ns2 = 4, 8
    slab1, slab2 = 0, ns1 * cp * Cin
    tab = (H.ConvParam * 2)()
    tab[0] = H.ConvParam(o_w1, -1, -1, -1, -1, wf1, wd1, 0, slab1, 0, Cout, Cin, 1, cp, ns1, 0, 0, 0)
    tab[1] = H.ConvParam(o_w2, o_g, o_b, 0, C2o, wf2, wd2, 2 * cp, slab2, 4 * ns1 * cp, C2o, C2i, 9, C2o, ns2, 0, 0, 0)
    tdev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda()
    warena = torch.zeros(wbytes, dtype=torch.uint8, device="cuda")
    ss = torch.zeros(2 * cp + 2 * C2o, device="cuda")
    H.call("dcf_weight_prep", dtype, tdev, 2, params, buffers, warena, ss, 1e-5, H.stream_ptr())
    td = TORCH_DT[dtype]
    scale = gamma / torch.sqrt(var + 1e-5)
    got_wf2 = warena[wf2:wf2 + w2.numel() * es].view(td).float().cpu().view(C2o, 9 * C2i)
    ref_wf2 = (w2.reshape(C2o, -1) * scale.view(-1, 1))
    assert rel_err(got_wf2, q(ref_wf2, dtype)) < (1e-6 if dtype == 0 else 1e-2)
    got_wd2 = warena[wd2:wd2 + w2.numel() * es].view(td).float().cpu().view(C2i, 9, C2o)
    assert torch.equal(got_wd2.permute(2, 1, 0).reshape(C2o, -1), got_wf2)
    got_wf1 = warena[wf1:wf1 + cp * Cin * es].view(td).float().cpu().view(cp, Cin)
    assert torch.equal(got_wf1[:Cout], q(w1.reshape(Cout, Cin), dtype)) and float(got_wf1[Cout:].abs().max()) == 0.0
    ssc = ss.cpu()
    assert torch.allclose(ssc[2 * cp:2 * cp + C2o], scale, rtol=1e-6)
    assert torch.allclose(ssc[2 * cp + C2o:], beta - mean * scale, rtol=1e-5, atol=1e-6)
    assert torch.equal(ssc[:Cout], torch.ones(Cout)) and float(ssc[Cout:2 * cp].abs().max()) == 0.0
    # finalize: random slabs + gsum -> dW, dgamma, dbeta vs the analytic chain rule
    slabs = torch.cat((rnd((ns1, cp, Cin), 22).reshape(-1), rnd((ns2, C2o, 9 * C2i), 23).reshape(-1))).cuda()
    gsum = rnd((4 * ns1 * cp + 4 * ns2 * C2o,), 24).cuda()       # [4*nsplit][cout_pad] per layer
    grads = torch.zeros_like(params)
    H.call("dcf_wgrad_finalize", tdev, 2, max(Cout, C2o), params, buffers, ss, slabs, gsum, grads, 1e-5, H.stream_ptr())
    gr = grads.cpu()
    G1 = slabs.cpu()[:ns1 * cp * Cin].view(ns1, cp, Cin).sum(0)[:Cout]
    G2 = slabs.cpu()[ns1 * cp * Cin:].view(ns2, C2o, 9 * C2i).sum(0)
    assert torch.allclose(gr[o_w1:o_w1 + w1.numel()].view(Cout, Cin), G1, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gr[o_w2:o_g].view(C2o, -1), G2 * scale.view(-1, 1), rtol=1e-5, atol=1e-6)
    dbeta = gsum.cpu()[4 * ns1 * cp:].view(4 * ns2, C2o).sum(0)
    dgamma = ((w2.reshape(C2o, -1) * G2).sum(1) - mean * dbeta) / torch.sqrt(var + 1e-5)
    assert torch.allclose(gr[o_b:o_b + C2o], dbeta, rtol=1e-5, atol=1e-6)
    assert torch.allclose(gr[o_g:o_g + C2o], dgamma, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_bn_train_kernels(dtype):
    """Train-mode BatchNorm2d forward (+ running-stat update, residual, ReLU) and backward against F.batch_norm."""
    ops = pkg("ops")
    C, B, Hh, W = 96, 2, 9, 7
    x = q(rnd((B, C, Hh, W), 51, -2.0, 2.0) + 0.3, dtype).requires_grad_(True)
    res = q(rnd((B, C, Hh, W), 52), dtype)
    gamma, beta = rnd((C,), 53, 0.5, 1.5).requires_grad_(True), rnd((C,), 54, -0.2, 0.2).requires_grad_(True)
    rm, rv = rnd((C,), 55, -0.1, 0.1), rnd((C,), 56, 0.8, 1.2)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = torch.relu(F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5) + res)
    ws = ops.bn_workspace(C, "cuda")
    rmd, rvd = rm.cuda(), rv.cuda()
    y, mean, invstd = ops.bn_train_fwd(dtype, to_dev(x.detach(), dtype), gamma.detach().cuda(), beta.detach().cuda(), to_dev(res, dtype),
                                       rmd, rvd, True, ws)
    assert rel_err(from_dev(y), ref.detach()) < (1e-5 if dtype == 0 else 1e-2)
    assert torch.allclose(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rvd.cpu(), rv_ref, rtol=1e-4, atol=1e-6)
    g = q(rnd((B, C, Hh, W), 57), dtype) * (ref.detach() > 0)
    ref.backward(g)
    dgam, dbet = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx = ops.bn_train_bwd(dtype, to_dev(g, dtype), to_dev(x.detach(), dtype), mean, invstd, gamma.detach().cuda(), dgam, dbet, ws)
    tol = 1e-4 if dtype == 0 else 2e-2
    assert rel_err(from_dev(dx), x.grad) < tol
    assert rel_err(dgam.cpu(), gamma.grad) < tol and rel_err(dbet.cpu(), beta.grad) < tol
