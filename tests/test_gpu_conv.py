"""GPU parity: MFMA implicit-GEMM convolution (fwd / dgrad / wgrad) through the C ABI against
torch's fp32 CPU convolution.  fp32 path: 1e-4 relative; bf16 path: inputs pre-rounded to bf16,
fp32 accumulate, 1e-2 relative to the output's max (output rounding to bf16)."""
import pytest
import torch
import torch.nn.functional as F

from _util import TORCH_DT, from_dev, pkg, q, rel_err, rnd, to_dev

pytestmark = pytest.mark.gpu

TOL = {0: 2e-4, 1: 1.2e-2, 2: 2e-3}        # fp32 / bf16 (8 significant bits) / fp16 (11)

# (B, H, W, Cin, Cout, k, stride)  -- covers every tile dispatch (Cout%128, %64, %32; row bytes %128 or %64)
SHAPES = [
    (1, 16, 16, 32, 32, 3, 1),
    (2, 17, 13, 32, 64, 3, 2),
    (1, 12, 20, 64, 128, 3, 1),
    (2, 10, 10, 128, 192, 3, 2),
    (1, 9, 11, 96, 160, 3, 1),
    (1, 8, 8, 192, 256, 1, 1),
    (2, 14, 10, 64, 96, 1, 2),
    (1, 24, 16, 256, 192, 1, 1),
    (1, 40, 52, 32, 32, 3, 1),
    # 3x3/s1 layers wide enough for the LDS-DMA weight-gradient kernel (Wo + 2 >= 40 resp. 48)
    (2, 12, 40, 64, 64, 3, 1),
    (1, 9, 47, 96, 160, 3, 1),       # partial channel tiles on both operands
    (2, 7, 50, 128, 64, 3, 1),
    (1, 20, 46, 32, 64, 3, 1),       # 64 x 32 tile
    (2, 5, 61, 64, 32, 3, 1),        # 32 x 64 tile
    (3, 33, 75, 64, 64, 3, 1),       # many stages per wave, ranges crossing image boundaries
    # row-sharing forward / dgrad kernel at awkward sizes: tail tiles with whole waves out of range (their stores are issued
    # all the same: the next tile's counted waits rely on the count), odd widths, three frames
    (2, 37, 61, 64, 64, 3, 1),
    (1, 75, 83, 128, 128, 3, 1),
    (3, 41, 43, 192, 128, 3, 1),
    # strip-walking weight-gradient kernel of the 32 -> 32 layers (conv_wgv.hip): widths that are no multiple of the 32-column
    # strip, heights that are no multiple of the row range, fewer units than a workgroup has waves, several frames
    (2, 37, 45, 32, 32, 3, 1),
    (3, 9, 100, 32, 32, 3, 1),
    (1, 8, 8, 32, 32, 3, 1),
    (2, 130, 33, 32, 32, 3, 1),
    # shared-staging weight-gradient kernel: 128-multiple channels (2 x 2 quadrants) and 192 x 192 (3 x 1)
    (2, 9, 44, 128, 128, 3, 1),
    (3, 21, 41, 256, 128, 3, 1),     # two input-channel tiles; pixel ranges crossing frames
    (1, 13, 50, 192, 192, 3, 1),
    (2, 30, 38, 192, 192, 3, 1),
    # stride-2 3x3 layers (generic weight-gradient kernel, parity-class input gradient): pixel ranges crossing rows and frames,
    # partial channel tiles on both operands, odd sizes
    (3, 23, 37, 64, 128, 3, 2),
    (1, 31, 29, 96, 160, 3, 2),
]


def _mk(shape, dtype, seed):
    B, Hh, W, Cin, Cout, k, s = shape
    x = q(rnd((B, Cin, Hh, W), seed), dtype)
    w = q(rnd((Cout, Cin, k, k), seed + 1, -0.2, 0.2), dtype)
    return x, w


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_fwd(shape, dtype):
    ops = pkg("ops")
    B, Hh, W, Cin, Cout, k, s = shape
    pad = k // 2
    x, w = _mk(shape, dtype, 11)
    shift = rnd((Cout,), 5)
    ref_lin = F.conv2d(x, w, None, s, pad)
    res = q(rnd(tuple(ref_lin.shape), 6), dtype)
    xd, wd = to_dev(x, dtype), to_dev(w, dtype)   # weights [Cout][kh][kw][Cin]
    # plain
    y = ops.conv2d_fwd(dtype, xd, wd, None, None, k, k, s, pad, False, Cout)
    e = rel_err(from_dev(y), ref_lin)
    assert e < TOL[dtype], "plain conv rel err %g" % e
    # fused epilogue: shift + residual + relu
    y2 = ops.conv2d_fwd(dtype, xd, wd, shift.cuda(), to_dev(res, dtype), k, k, s, pad, True, Cout)
    ref2 = torch.relu(ref_lin + shift.view(1, -1, 1, 1) + res)
    e2 = rel_err(from_dev(y2), ref2)
    assert e2 < TOL[dtype], "fused conv rel err %g" % e2
    assert float(from_dev(y2).min()) >= 0.0


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_dgrad(shape, dtype):
    ops = pkg("ops")
    B, Hh, W, Cin, Cout, k, s = shape
    pad = k // 2
    x, w = _mk(shape, dtype, 21)
    x.requires_grad_(True)
    y = F.conv2d(x, w, None, s, pad)
    gy = q(rnd(tuple(y.shape), 22), dtype)
    y.backward(gy)
    wt = w.permute(1, 2, 3, 0).contiguous().cuda()          # [Cin][kh][kw][Cout]
    wt = wt.to(TORCH_DT[dtype])
    gx = ops.conv2d_dgrad(dtype, to_dev(gy, dtype), wt, None, (B, Hh, W, Cin), k, k, s, pad)
    e = rel_err(from_dev(gx), x.grad)
    assert e < TOL[dtype], "dgrad rel err %g" % e
    res = q(rnd((B, Cin, Hh, W), 23), dtype)
    gx2 = ops.conv2d_dgrad(dtype, to_dev(gy, dtype), wt, to_dev(res, dtype), (B, Hh, W, Cin), k, k, s, pad)
    e2 = rel_err(from_dev(gx2), x.grad + res)
    assert e2 < TOL[dtype], "dgrad+res rel err %g" % e2
    # fused ReLU backward of the producer: gx *= (mask > 0)
    mask = q(rnd((B, Cin, Hh, W), 24), dtype)
    gx3 = ops.conv2d_dgrad(dtype, to_dev(gy, dtype), wt, to_dev(res, dtype), (B, Hh, W, Cin), k, k, s, pad, to_dev(mask, dtype))
    ref3 = (x.grad + res) * (mask > 0)
    got3 = from_dev(gx3)
    assert rel_err(got3, ref3) < TOL[dtype]
    assert float((got3 * (mask <= 0)).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [0, 1, 2])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_wgrad(shape, dtype):
    ops = pkg("ops")
    B, Hh, W, Cin, Cout, k, s = shape
    pad = k // 2
    x, w = _mk(shape, dtype, 31)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, s, pad)
    gy = q(rnd(tuple(y.shape), 32), dtype)
    y.backward(gy)
    Ho, Wo = y.shape[-2:]
    ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, k, k, s)
    assert ns >= 1
    slabs = torch.full((ns, Cout, k, k, Cin), float("nan"), device="cuda")
    ops.conv2d_wgrad(dtype, to_dev(x, dtype), to_dev(gy, dtype), slabs, ns, k, k, s, pad)
    G = slabs.sum(0).cpu().permute(0, 3, 1, 2)                  # -> [Cout,Cin,kh,kw]
    assert torch.isfinite(G).all(), "wgrad left unwritten slab entries"
    e = rel_err(G, w.grad)
    assert e < (2e-4 if dtype == 0 else 2e-3), "wgrad rel err %g" % e
    # fixed-order slabs: bitwise reproducible
    slabs2 = torch.zeros_like(slabs)
    gsum = torch.full((4 * ns, Cout), float("nan"), device="cuda")
    ops.conv2d_wgrad(dtype, to_dev(x, dtype), to_dev(gy, dtype), slabs2, ns, k, k, s, pad, gsum)
    assert torch.equal(slabs, slabs2)
    # per-split column sums of gy (dbeta of a folded BN) per wave
    want = gy.sum((0, 2, 3))
    got = gsum.sum(0).cpu()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) < (1e-4 if dtype == 0 else 1e-3) * float(gy.abs().sum((0, 2, 3)).max())


@pytest.mark.parametrize("dtype", [0, 1, 2])
def test_stem7x7(dtype):
    """7x7/2 RGB stem on the NHWC4+halo image against conv2d(x/255, w, stride 2, pad 3)."""
    ops, det = pkg("ops"), pkg("detfill")
    B, Hh, W, Cout = 2, 38, 50, 64
    img = torch.from_numpy(det.synthetic_image(Hh, W, 3)).unsqueeze(0).repeat(B, 1, 1, 1).contiguous()
    img[1] = torch.flip(img[1], dims=[2])
    w = q(rnd((Cout, 3, 7, 7), 41, -0.2, 0.2), dtype)
    xf = q(img.float() / 255.0, dtype)
    ref = F.conv2d(xf, w, None, 2, 3)
    w8 = torch.zeros(Cout, 7, 8, 4)
    w8[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    wd = w8.cuda().to(TORCH_DT[dtype])
    img4 = ops.image_to_nhwc4(img.cuda(), dtype)
    y = ops.stem7x7_fwd(dtype, img4, wd, None, False, Cout, Hh, W)
    e = rel_err(from_dev(y), ref)
    assert e < TOL[dtype], "stem fwd rel err %g" % e
    # wgrad
    wv = w.clone().requires_grad_(True)
    out = F.conv2d(xf, wv, None, 2, 3)
    gy = q(rnd(tuple(out.shape), 42), dtype)
    out.backward(gy)
    ns = ops.conv2d_wgrad_splits(B, out.shape[2], out.shape[3], 32, Cout, 7, 1, 2)
    slabs = torch.zeros((ns, Cout, 7, 8, 4), device="cuda")
    ops.stem7x7_wgrad(dtype, img4, to_dev(gy, dtype), slabs, ns, Hh, W)
    G = slabs.sum(0).cpu()[:, :, :7, :3].permute(0, 3, 1, 2)
    e = rel_err(G, wv.grad)
    assert e < (2e-4 if dtype == 0 else 3e-3), "stem wgrad rel err %g" % e


def test_conv_full_size_layer_linearity():
    """BASELINE-size check without a CPU reference: conv is linear, so conv(a*x1 + x2) == a*conv(x1) + conv(x2)
    on a full KITTI-scale layer1 tensor (fp32 path), and bf16 agrees with fp32 to bf16 precision."""
    ops = pkg("ops")
    B, Hh, W, C = 1, 704, 800, 32
    g = torch.Generator(device="cuda").manual_seed(0)
    x1 = torch.rand((B, Hh, W, C), device="cuda", generator=g) - 0.5
    x2 = torch.rand((B, Hh, W, C), device="cuda", generator=g) - 0.5
    w = (torch.rand((C, 3, 3, C), device="cuda", generator=g) - 0.5) * 0.2
    y1 = ops.conv2d_fwd(0, x1, w, None, None, 3, 3, 1, 1, False, C)
    y2 = ops.conv2d_fwd(0, x2, w, None, None, 3, 3, 1, 1, False, C)
    y3 = ops.conv2d_fwd(0, (2.0 * x1 + x2).contiguous(), w, None, None, 3, 3, 1, 1, False, C)
    assert float((y3 - (2.0 * y1 + y2)).abs().max()) < 1e-4
    yb = ops.conv2d_fwd(1, x1.bfloat16(), w.bfloat16(), None, None, 3, 3, 1, 1, False, C).float()
    yr = ops.conv2d_fwd(0, x1.bfloat16().float(), w.bfloat16().float(), None, None, 3, 3, 1, 1, False, C)
    assert float((yb - yr).abs().max() / yr.abs().max()) < 1e-2


@pytest.mark.parametrize("dtype", [1, 2])
def test_wgrad_group_equals_single_launches(dtype):
    """dcf_conv2d_wgrad_group (layers collected and issued kernel class by kernel class, up to 32 per launch) writes
    bitwise the same slabs / dbeta sums as one dcf_conv2d_wgrad per layer: row-sharing LDS-DMA layers (several, so that
    workgroup offsets matter), generic stride-2 / 1x1 layers, and a 32-channel layer that goes out on its own."""
    import ctypes
    ops, H = pkg("ops"), pkg("_hip")
    shapes = [(2, 12, 40, 64, 64, 3, 1), (1, 9, 47, 128, 64, 3, 1), (2, 7, 50, 64, 128, 3, 1), (2, 17, 13, 64, 64, 3, 2),
              (1, 24, 16, 256, 192, 1, 1), (2, 10, 10, 128, 192, 3, 2), (1, 40, 52, 32, 32, 3, 1), (1, 8, 8, 192, 256, 1, 1),
              (2, 9, 44, 128, 128, 3, 1), (1, 30, 60, 128, 256, 3, 1), (1, 13, 50, 192, 192, 3, 1), (2, 30, 38, 192, 192, 3, 1)]
    keep, items, want = [], [], []
    for i, (B, Hh, W, Cin, Cout, k, s) in enumerate(shapes):
        pad = k // 2
        Ho, Wo = (Hh + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = to_dev(q(rnd((B, Cin, Hh, W), 300 + i), dtype), dtype)
        gy = to_dev(q(rnd((B, Cout, Ho, Wo), 400 + i), dtype), dtype)
        ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, k, k, s)
        ref = torch.full((ns, Cout, k, k, Cin), float("nan"), device="cuda")
        refs = torch.full((4 * ns, Cout), float("nan"), device="cuda")
        ops.conv2d_wgrad(dtype, x, gy, ref, ns, k, k, s, pad, refs)
        got = torch.full_like(ref, float("nan"))
        gots = torch.full_like(refs, float("nan"))
        keep += [x, gy, got, gots]
        want.append((ref, refs, got, gots))
        items.append(H.WgradItem(dtype, ns, x.data_ptr(), gy.data_ptr(), got.data_ptr(), gots.data_ptr(), B, Hh, W, Cin, Cout, k, k, s, pad, 0))
    arr = (H.WgradItem * len(items))(*items)
    H.call("dcf_conv2d_wgrad_group", ctypes.addressof(arr), len(items), H.stream_ptr())
    torch.cuda.synchronize()
    for ref, refs, got, gots in want:
        assert torch.equal(ref, got) and torch.equal(refs, gots)


# ---- launches with MORE than 512 workgroups: the register-staged k_conv_igemm<..., DB=false> instantiations (and the
# big-tile kernels that replace them) that the cfg2 bench actually runs -- the small shapes above all dispatch to the
# <= 512-workgroup LDS-DMA / double-buffered variants.  (B, H, W, Cin, Cout, k, stride, what it exercises)
BIG_SHAPES = [
    (1, 352, 400, 128, 128, 3, 1),   # 128x128 tiles: 1100 workgroups (stage-3 body at 4x the pixels)
    (1, 352, 400, 64, 64, 3, 1),     # 64 ch x 128 px tiles: 1100 workgroups (stage-2 body)
    (1, 704, 800, 32, 32, 3, 1),     # 32 ch x 128 px tiles: 4400 workgroups, 64-byte rows (stage-1 body, full size)
    (1, 704, 800, 64, 128, 3, 2),    # stride 2: forward 1100 workgroups of 128x128; dgrad = parity classes, 4400 tiles of 64x128
    (1, 353, 399, 128, 192, 3, 2),   # stride 2, odd sizes: unequal parity classes; 192 = 64-channel tiles
    (2, 176, 200, 192, 192, 3, 1),   # conv3 / FPN shape: 192 channels -> 64 ch x 128 px tiles, batch boundary inside tiles
    (1, 352, 400, 128, 192, 1, 1),   # 1x1 (latconv2 shape), 128 -> 192
    (1, 301, 397, 64, 64, 3, 1),     # row-sharing kernel, persistent workgroups over three rounds of tiles, odd sizes
    (2, 203, 199, 128, 128, 3, 1),   # same on the 128-channel kind (five / four position tiles per wave)
    # ResNet-50 Bottleneck 1x1 convolutions at cfg4's sizes (batch 4, 1242x375 image)
    (4, 94, 311, 256, 64, 1, 1),     # layer1 conv1: 117 k pixels, 914 x 1 tiles; dgrad widens 64 -> 256
    (4, 24, 78, 1024, 256, 1, 1),    # layer3 conv1: K = 1024
    (4, 12, 39, 512, 2048, 1, 1),    # layer4 conv3 shape class (wide output: 16-32 channel tiles per pixel tile)
    (4, 47, 156, 256, 512, 1, 2),    # layer2 downsample: 1x1 / stride 2
]


def _assert_quantised_close(got, ref, dtype, what):
    """Quantisation-aware comparison: the device accumulates in fp32 and rounds ONCE to the storage type, so every
    element must be within one unit in the last place of the storage type of the fp32 reference (plus the fp32
    accumulation-order slack, relative to the tensor's scale) -- far tighter than a max-relative bound."""
    ulp = {0: 2.0 ** -22, 1: 2.0 ** -8, 2: 2.0 ** -11}[dtype]
    slack = 2e-5 * float(ref.abs().max())
    bad = (got - ref).abs() > ulp * ref.abs() + slack
    assert not bool(bad.any()), "%s: %d of %d elements off by more than one storage ulp, worst %g at ref %g" % (
        what, int(bad.sum()), bad.numel(), float((got - ref).abs()[bad].max()), float(ref[bad].abs().max()))


@pytest.mark.parametrize("dtype", [1, 2, 0])
@pytest.mark.parametrize("shape", BIG_SHAPES)
def test_conv_fwd_big_launch(shape, dtype):
    ops = pkg("ops")
    B, Hh, W, Cin, Cout, k, s = shape
    pad = k // 2
    x, w = _mk(shape, dtype, 51)
    shift = rnd((Cout,), 52)
    ref_lin = F.conv2d(x, w, None, s, pad)
    res = q(rnd(tuple(ref_lin.shape), 53), dtype)
    xd, wd = to_dev(x, dtype), to_dev(w, dtype)
    y = ops.conv2d_fwd(dtype, xd, wd, None, None, k, k, s, pad, False, Cout)
    _assert_quantised_close(from_dev(y), ref_lin, dtype, "plain")
    y2 = ops.conv2d_fwd(dtype, xd, wd, shift.cuda(), to_dev(res, dtype), k, k, s, pad, True, Cout)
    _assert_quantised_close(from_dev(y2), torch.relu(ref_lin + shift.view(1, -1, 1, 1) + res), dtype, "fused")


@pytest.mark.parametrize("dtype", [1, 2, 0])
@pytest.mark.parametrize("shape", BIG_SHAPES)
def test_conv_dgrad_big_launch(shape, dtype):
    ops = pkg("ops")
    B, Hh, W, Cin, Cout, k, s = shape
    pad = k // 2
    x, w = _mk(shape, dtype, 61)
    x.requires_grad_(True)
    y = F.conv2d(x, w, None, s, pad)
    gy = q(rnd(tuple(y.shape), 62), dtype)
    y.backward(gy)
    wt = w.permute(1, 2, 3, 0).contiguous().cuda().to(TORCH_DT[dtype])          # [Cin][kh][kw][Cout]
    gyd = to_dev(gy, dtype)
    gx = ops.conv2d_dgrad(dtype, gyd, wt, None, (B, Hh, W, Cin), k, k, s, pad)
    _assert_quantised_close(from_dev(gx), x.grad, dtype, "dgrad")
    res = q(rnd((B, Cin, Hh, W), 63), dtype)
    mask = q(rnd((B, Cin, Hh, W), 64), dtype)
    gx3 = ops.conv2d_dgrad(dtype, gyd, wt, to_dev(res, dtype), (B, Hh, W, Cin), k, k, s, pad, to_dev(mask, dtype))
    got3 = from_dev(gx3)
    _assert_quantised_close(got3, (x.grad + res) * (mask > 0), dtype, "dgrad+res+mask")
    assert float((got3 * (mask <= 0)).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [1, 0, 2])
@pytest.mark.parametrize("shape", [(2, 24, 40, 64, 64), (1, 33, 35, 128, 128), (2, 352, 400, 64, 64), (3, 11, 13, 256, 192)])
def test_conv_fwd_rowscale(shape, dtype):
    """dcf_conv2d_fwd_rowscale: y = conv1x1(x, w) + cnt[m] * b[c] + res in one epilogue (the fusion site's fc2 under the neighbour sum,
    reference model.py:216-219), against the fp32 statement, incl. the launch of the stride-2 site's size."""
    ops, H = pkg("ops"), pkg("_hip")
    B, Hh, W, Cin, Cout = shape
    x, w = _mk((B, Hh, W, Cin, Cout, 1, 1), dtype, 81)
    b = rnd((Cout,), 82)
    cnt = torch.randint(0, 4, (B, Hh * W), generator=torch.Generator().manual_seed(83)).float()
    res = q(rnd((B, Cout, Hh, W), 84), dtype)
    ref = F.conv2d(x, w) + res + cnt.view(B, 1, Hh, W) * b.view(1, -1, 1, 1)
    y = ops.conv2d_fwd_rowscale(dtype, to_dev(x, dtype), to_dev(w, dtype), b.cuda(), cnt.cuda(), to_dev(res, dtype), False, Cout)
    _assert_quantised_close(from_dev(y), ref, dtype, "conv + cnt*b + res")
    with pytest.raises(H.DcfError):
        ops.conv2d_fwd_rowscale(dtype, to_dev(x, dtype), to_dev(w, dtype), b.cuda(), cnt.cuda()[:, :-1].contiguous(), None, False, Cout)


HALFRES_SHAPES = [
    (2, 17, 13, 32, 64, 3, 2),       # odd sizes: the even sub-grid is the larger class
    (2, 10, 10, 128, 192, 3, 2),
    (3, 24, 40, 64, 128, 3, 2),
    (1, 704, 800, 32, 64, 3, 2),     # stage-2 entry block at full size (register-staged kernel, 64-byte pixels)
    (1, 353, 399, 128, 192, 3, 2),   # odd sizes, unequal parity classes
]


@pytest.mark.parametrize("dtype", [1, 0, 2])
@pytest.mark.parametrize("shape", HALFRES_SHAPES)
def test_conv_dgrad_halfres_residual(shape, dtype):
    """dcf_conv2d_dgrad_halfres: the residual given on the (2i, 2j) sub-grid is bit-for-bit the full-resolution residual that
    is zero elsewhere (same kernel, same accumulation order), also next to a full-resolution residual and a mask."""
    ops = pkg("ops")
    H = pkg("_hip")
    B, Hh, W, Cin, Cout, k, s = shape
    pad = k // 2
    x, w = _mk(shape, dtype, 71)
    x.requires_grad_(True)
    y = F.conv2d(x, w, None, s, pad)
    gy = q(rnd(tuple(y.shape), 72), dtype)
    y.backward(gy)
    wt = w.permute(1, 2, 3, 0).contiguous().cuda().to(TORCH_DT[dtype])
    gyd = to_dev(gy, dtype)
    Hq, Wq = (Hh + 1) // 2, (W + 1) // 2
    rq = q(rnd((B, Cin, Hq, Wq), 73), dtype)
    full = torch.zeros((B, Cin, Hh, W))
    full[:, :, ::2, ::2] = rq
    res = q(rnd((B, Cin, Hh, W), 74), dtype)
    mask = q(rnd((B, Cin, Hh, W), 75), dtype)
    a = ops.conv2d_dgrad_halfres(dtype, gyd, wt, None, to_dev(rq, dtype), (B, Hh, W, Cin), k, k, s, pad, to_dev(mask, dtype))
    b = ops.conv2d_dgrad(dtype, gyd, wt, to_dev(full, dtype), (B, Hh, W, Cin), k, k, s, pad, to_dev(mask, dtype))
    assert torch.equal(a, b)
    _assert_quantised_close(from_dev(a), (x.grad + full) * (mask > 0), dtype, "dgrad+resq+mask")
    c = ops.conv2d_dgrad_halfres(dtype, gyd, wt, to_dev(res, dtype), to_dev(rq, dtype), (B, Hh, W, Cin), k, k, s, pad)
    _assert_quantised_close(from_dev(c), x.grad + res + full, dtype, "dgrad+res+resq")
    with pytest.raises(H.DcfError):      # wrong grid
        ops.conv2d_dgrad_halfres(dtype, gyd, wt, None, to_dev(res, dtype), (B, Hh, W, Cin), k, k, s, pad)


def test_conv_dgrad_halfres_rejects_layers_without_parity_classes():
    ops = pkg("ops")
    H = pkg("_hip")
    gy = torch.zeros((1, 8, 8, 32), dtype=torch.bfloat16, device="cuda")
    wt = torch.zeros((32, 3, 3, 32), dtype=torch.bfloat16, device="cuda")
    rq = torch.zeros((1, 4, 4, 32), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(H.DcfError):
        ops.conv2d_dgrad_halfres(1, gy, wt, None, rq, (1, 8, 8, 32), 3, 3, 1, 1)


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("shape", [BIG_SHAPES[0], BIG_SHAPES[5], BIG_SHAPES[3], BIG_SHAPES[9], BIG_SHAPES[10], BIG_SHAPES[12], BIG_SHAPES[2],
                                   (2, 301, 397, 32, 32, 3, 1)])
def test_conv_wgrad_big_launch(shape, dtype):
    """Weight gradient at bench-size pixel counts (many pixel ranges per layer, ranges crossing image rows / frames)."""
    ops = pkg("ops")
    B, Hh, W, Cin, Cout, k, s = shape
    pad = k // 2
    x, w = _mk(shape, dtype, 71)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, s, pad)
    gy = q(rnd(tuple(y.shape), 72), dtype)
    y.backward(gy)
    Ho, Wo = y.shape[-2:]
    ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, k, k, s)
    slabs = torch.full((ns, Cout, k, k, Cin), float("nan"), device="cuda")
    gsum = torch.full((4 * ns, Cout), float("nan"), device="cuda")
    ops.conv2d_wgrad(dtype, to_dev(x, dtype), to_dev(gy, dtype), slabs, ns, k, k, s, pad, gsum)
    G = slabs.sum(0).cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(G).all()
    e = rel_err(G, w.grad)
    assert e < 2e-4, "wgrad rel err %g" % e                       # fp32 accumulation of exactly representable products
    want = gy.sum((0, 2, 3))
    assert float((gsum.sum(0).cpu() - want).abs().max()) < 2e-4 * float(gy.abs().sum((0, 2, 3)).max())


# ---- the small-M kind of the row-sharing kernel with eight consumer + eight loader waves (option RS_L16, the default for that kind
# since round 5): every consumer wave does exactly what it does in the 8-wave form, the loader waves issue the same DMA pieces into the
# same LDS slots -- so the results must be BIT-identical, also over persistent workgroups that walk several tiles (RS_KIND / RS_NPT
# force the small-M kind onto launches of more tiles than workgroups) and for the 128- / 192-channel kinds that the option can reach.
@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("case", [((2, 44, 50, 256, 256), None, None), ((2, 24, 78, 256, 256), None, None), ((2, 12, 39, 512, 512), None, None),
                                  ((6, 44, 50, 256, 256), 2, 1), ((3, 88, 100, 64, 64), 2, 3), ((2, 88, 100, 192, 192), None, None)])
def test_conv_rs_sixteen_wave_form_equals_eight_wave_form(case, dtype):
    ops, H = pkg("ops"), pkg("_hip")
    (B, Hh, W, Cin, Cout), kind, npt = case
    shape = (B, Hh, W, Cin, Cout, 3, 1)
    x, w = _mk(shape, dtype, 171)
    shift = rnd((Cout,), 172).cuda()
    res = to_dev(q(rnd((B, Cout, Hh, W), 173), dtype), dtype)
    xd, wd = to_dev(x, dtype), to_dev(w, dtype)
    outs = []
    try:
        if kind is not None:
            H.set_option("RS_KIND", kind); H.set_option("RS_NPT", npt)
        for l16 in (0, 7):
            H.set_option("RS_L16", l16)
            y = ops.conv2d_fwd(dtype, xd, wd, shift, res, 3, 3, 1, 1, True, Cout)
            outs.append(y.clone())
    finally:
        H.set_option("RS_L16", None); H.set_option("RS_KIND", None); H.set_option("RS_NPT", None)
    assert torch.equal(outs[0], outs[1])
    ref = torch.relu(F.conv2d(x, w, None, 1, 1) + shift.cpu().view(1, -1, 1, 1) + from_dev(res))
    _assert_quantised_close(from_dev(outs[1]), ref, dtype, "16-wave form")


# ---- shared-staging weight-gradient kernel of the 1x1 (any stride) and 3x3 / stride-2 layers (conv_wg1.hip).  The product sends it
# layers of >= 2048 output pixels with 64-multiple channel counts (BIG_SHAPES above: the stride-2 stage heads, the ResNet-50 1x1s);
# here it is FORCED onto small, awkward shapes (option WGRAD1S_MIN_PIXELS = 1): output rows shorter than a 32-pixel stage, ranges
# crossing rows and frames, half-empty and several channel tiles on either operand, fewer pixels than one stage per group.
WG1_SHAPES = [
    (2, 14, 10, 64, 128, 1, 2),      # 1x1 / stride 2, output rows of 5 pixels: a stage spans 6-7 rows
    (1, 24, 16, 256, 192, 1, 1),     # two input-channel tiles, 1.5 output-channel tiles
    (2, 9, 7, 320, 64, 1, 1),        # 2.5 input-channel tiles, half an output-channel tile
    (3, 23, 37, 64, 128, 3, 2),      # 3x3 / stride 2: nine taps, image borders on every side, frames inside ranges
    (1, 31, 29, 128, 192, 3, 2),     # odd sizes, partial output-channel tile
    (1, 5, 5, 64, 64, 1, 1),         # 25 pixels: less than one stage per group
    (4, 47, 156, 256, 512, 1, 2),    # cfg4 layer2 downsample (the product's own route)
    (2, 88, 100, 192, 256, 3, 2),    # cfg2 stage-5 head (the product's own route)
    (2, 20, 90, 64, 128, 3, 2),      # 3x3 / stride 2: 64 input channels (no second x sub-tile), rows of 45 pixels, frames inside ranges
    (1, 9, 79, 128, 192, 3, 2),      # odd width: the last tap column is the image's last column
    (3, 7, 100, 192, 64, 3, 2),      # 1.5 input-channel tiles, half an output-channel tile, odd height
    (1, 353, 399, 128, 192, 3, 2),   # odd sizes, many stages per group
]


@pytest.fixture
def force_wg1():
    H = pkg("_hip")
    H.set_option("WGRAD1S_MIN_PIXELS", 1)
    yield
    H.set_option("WGRAD1S_MIN_PIXELS", None)


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("shape", WG1_SHAPES)
def test_conv_wgrad_shared_staging_1x1_and_stride2(shape, dtype, force_wg1):
    ops, H = pkg("ops"), pkg("_hip")
    B, Hh, W, Cin, Cout, k, s = shape
    pad = k // 2
    x, w = _mk(shape, dtype, 131)
    w.requires_grad_(True)
    y = F.conv2d(x, w, None, s, pad)
    gy = q(rnd(tuple(y.shape), 132), dtype)
    y.backward(gy)
    Ho, Wo = y.shape[-2:]
    xd, gd = to_dev(x, dtype), to_dev(gy, dtype)
    ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, k, k, s)
    H.call("dcf_prof_reset"); H.call("dcf_prof_enable", 1)
    slabs = torch.full((ns, Cout, k, k, Cin), float("nan"), device="cuda")
    gsum = torch.full((4 * ns, Cout), float("nan"), device="cuda")
    ops.conv2d_wgrad(dtype, xd, gd, slabs, ns, k, k, s, pad, gsum)
    torch.cuda.synchronize(); H.call("dcf_prof_enable", 0)
    assert any(name.startswith("conv_wgrad1s_grp") for name in H.prof_read()), "the launch did not take conv_wg1.hip"
    G = slabs.sum(0).cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(G).all(), "unwritten slab entries"
    assert rel_err(G, w.grad) < 2e-4                              # fp32 accumulation of exactly representable products
    want = gy.sum((0, 2, 3))
    assert float((gsum.sum(0).cpu() - want).abs().max()) < 2e-4 * float(gy.abs().sum((0, 2, 3)).max())
    slabs2 = torch.zeros_like(slabs)                              # fixed-order reduction: bitwise reproducible
    ops.conv2d_wgrad(dtype, xd, gd, slabs2, ns, k, k, s, pad)
    assert torch.equal(slabs, slabs2)
    # the generic kernel on the same split count (option WGRAD1S = 0): same sums up to the fp32 summation order
    H.set_option("WGRAD1S", 0)
    try:
        slabs3 = torch.full_like(slabs, float("nan"))
        ops.conv2d_wgrad(dtype, xd, gd, slabs3, ns, k, k, s, pad)
    finally:
        H.set_option("WGRAD1S", None)
    assert rel_err(slabs3.sum(0), slabs.sum(0)) < 2e-5


def test_wgrad_group_with_shared_staging_layers_equals_single_launches(force_wg1):
    """The conv_wg1.hip layers of a grouped launch (one launch for all of them, longest first, XCD rotation per layer) write bitwise
    what their single launches write."""
    import ctypes
    ops, H = pkg("ops"), pkg("_hip")
    dtype = 1
    keep, items, want = [], [], []
    for i, (B, Hh, W, Cin, Cout, k, s) in enumerate(WG1_SHAPES[:6] + [(2, 12, 40, 64, 64, 3, 1)]):
        pad = k // 2
        Ho, Wo = (Hh + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = to_dev(q(rnd((B, Cin, Hh, W), 500 + i), dtype), dtype)
        gy = to_dev(q(rnd((B, Cout, Ho, Wo), 600 + i), dtype), dtype)
        ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, k, k, s)
        ref = torch.full((ns, Cout, k, k, Cin), float("nan"), device="cuda")
        refs = torch.full((4 * ns, Cout), float("nan"), device="cuda")
        ops.conv2d_wgrad(dtype, x, gy, ref, ns, k, k, s, pad, refs)
        got, gots = torch.full_like(ref, float("nan")), torch.full_like(refs, float("nan"))
        keep += [x, gy, got, gots]
        want.append((ref, refs, got, gots))
        items.append(H.WgradItem(dtype, ns, x.data_ptr(), gy.data_ptr(), got.data_ptr(), gots.data_ptr(), B, Hh, W, Cin, Cout, k, k, s, pad, 0))
    arr = (H.WgradItem * len(items))(*items)
    H.call("dcf_conv2d_wgrad_group", ctypes.addressof(arr), len(items), H.stream_ptr())
    torch.cuda.synchronize()
    for ref, refs, got, gots in want:
        assert torch.equal(ref, got) and torch.equal(refs, gots)


# ---- spatial-tile streaming kernel (conv_sp.hip: Cin == Cout in {32, 64}, 3x3 / stride 1).  The product sends it the launches of
# >= 200 000 pixels (BIG_SHAPES above: 704x800x32, 352x400x64, 301x397x64); here it is FORCED onto small, awkward shapes
# (option CONV_SP_MIN_PIX = 0): partial tiles in both directions, tiles crossing nothing / everything, several frames, fewer
# tiles than workgroups, one-pixel-wide tails.
SP_SHAPES = [
    (1, 16, 16, 32, 32, 3, 1),
    (2, 9, 33, 32, 32, 3, 1),        # one pixel into the second tile column; 9 rows = one row into the second tile row
    (1, 40, 52, 32, 32, 3, 1),
    (3, 17, 31, 32, 32, 3, 1),
    (2, 37, 61, 64, 64, 3, 1),
    (3, 33, 75, 64, 64, 3, 1),
    (1, 8, 32, 64, 64, 3, 1),        # exactly one tile
    (1, 95, 129, 64, 64, 3, 1),      # more tiles than one round of workgroups per XCD chunk at 2 per CU
]


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("shape", SP_SHAPES)
def test_conv_sp_kernel_forced_on_small_shapes(shape, dtype):
    ops, H = pkg("ops"), pkg("_hip")
    B, Hh, W, Cin, Cout, k, s = shape
    x, w = _mk(shape, dtype, 81)
    shift = rnd((Cout,), 82)
    ref_lin = F.conv2d(x, w, None, 1, 1)
    res = q(rnd(tuple(ref_lin.shape), 83), dtype)
    xd, wd = to_dev(x, dtype), to_dev(w, dtype)
    xg = x.clone().requires_grad_(True)
    yr = F.conv2d(xg, w, None, 1, 1)
    gy = q(rnd(tuple(yr.shape), 84), dtype)
    yr.backward(gy)
    wt = w.permute(1, 2, 3, 0).contiguous().cuda().to(TORCH_DT[dtype])
    gres = q(rnd((B, Cin, Hh, W), 85), dtype)
    gmask = q(rnd((B, Cin, Hh, W), 86), dtype)
    try:
        H.set_option("CONV_SP_MIN_PIX", 0)
        H.call("dcf_prof_reset")
        H.call("dcf_prof_enable", 1)
        y = ops.conv2d_fwd(dtype, xd, wd, None, None, 3, 3, 1, 1, False, Cout)
        y2 = ops.conv2d_fwd(dtype, xd, wd, shift.cuda(), to_dev(res, dtype), 3, 3, 1, 1, True, Cout)
        gx = ops.conv2d_dgrad(dtype, to_dev(gy, dtype), wt, None, (B, Hh, W, Cin), 3, 3, 1, 1)
        gx3 = ops.conv2d_dgrad(dtype, to_dev(gy, dtype), wt, to_dev(gres, dtype), (B, Hh, W, Cin), 3, 3, 1, 1, to_dev(gmask, dtype))
        torch.cuda.synchronize()
        H.call("dcf_prof_enable", 0)
        names = H.prof_read()
    finally:
        H.call("dcf_prof_enable", 0)
        H.call("dcf_prof_reset")
        H.set_option("CONV_SP_MIN_PIX", None)
    assert any("<sp%d>" % Cin in n and n.startswith("conv_fwd") for n in names), sorted(names)
    assert any("<sp%d>" % Cin in n and n.startswith("conv_dgrad") for n in names), sorted(names)
    _assert_quantised_close(from_dev(y), ref_lin, dtype, "sp plain")
    _assert_quantised_close(from_dev(y2), torch.relu(ref_lin + shift.view(1, -1, 1, 1) + res), dtype, "sp fused")
    _assert_quantised_close(from_dev(gx), xg.grad, dtype, "sp dgrad")
    got3 = from_dev(gx3)
    _assert_quantised_close(got3, (xg.grad + gres) * (gmask > 0), dtype, "sp dgrad+res+mask")
    assert float((got3 * (gmask <= 0)).abs().max()) == 0.0


# ---- loader / consumer kernel over 2-D tiles (csrc/conv_lc.hip), forced on shapes that the automatic choice would leave to
# conv_rs.hip: odd widths / heights (tiles cut by the image border on every side), a width smaller than the smallest tile, one /
# two / three / four 64-channel chunks, 64-, 128-, 192- and 256-channel outputs (both tile kinds), several frames, enough tiles
# for the persistent loop to walk more than one per workgroup.  Forward with the fused epilogue, input gradient with residual +
# mask, against torch's fp32 convolution; and against conv_rs.hip on the same inputs (same products, another summation order).
LC_SHAPES = [
    (1, 16, 16, 64, 64),
    (2, 37, 61, 64, 64),
    (1, 75, 83, 128, 128),
    (3, 41, 43, 192, 128),
    (2, 9, 13, 256, 64),
    (1, 50, 101, 64, 192),
    (2, 23, 200, 128, 256),
    (1, 176, 200, 128, 128),          # 120 tiles of 6 x 50
    (4, 90, 102, 192, 192),           # > 256 tiles: the persistent loop
]


@pytest.mark.parametrize("dtype", [1, 2])
@pytest.mark.parametrize("shape", LC_SHAPES)
def test_conv_lc_fwd_dgrad(shape, dtype):
    ops, H = pkg("ops"), pkg("_hip")
    B, Hh, W, Cin, Cout = shape
    x, w = _mk((B, Hh, W, Cin, Cout, 3, 1), dtype, 41)
    shift = rnd((Cout,), 42)
    ref_lin = F.conv2d(x, w, None, 1, 1)
    res = q(rnd(tuple(ref_lin.shape), 43), dtype)
    xd, wd = to_dev(x, dtype), to_dev(w, dtype)
    try:
        H.set_option("CONV_LC", 2)
        y = ops.conv2d_fwd(dtype, xd, wd, None, None, 3, 3, 1, 1, False, Cout)
        y2 = ops.conv2d_fwd(dtype, xd, wd, shift.cuda(), to_dev(res, dtype), 3, 3, 1, 1, True, Cout)
        H.set_option("CONV_LC", 0)
        y_rs = ops.conv2d_fwd(dtype, xd, wd, None, None, 3, 3, 1, 1, False, Cout)
    finally:
        H.set_option("CONV_LC", None)
    assert rel_err(from_dev(y), ref_lin) < TOL[dtype]
    ref2 = torch.relu(ref_lin + shift.view(1, -1, 1, 1) + res)
    assert rel_err(from_dev(y2), ref2) < TOL[dtype]
    assert float(from_dev(y2).min()) >= 0.0
    # one unit in the last place of the 16-bit output at most, relative to the largest output
    assert rel_err(from_dev(y), from_dev(y_rs)) < (8e-3 if dtype == 1 else 1e-3)
    # input gradient: roles swap (reduction over Cout)
    xg = x.clone().requires_grad_(True)
    yy = F.conv2d(xg, w, None, 1, 1)
    gy = q(rnd(tuple(yy.shape), 44), dtype)
    yy.backward(gy)
    wt = w.permute(1, 2, 3, 0).contiguous().cuda().to(TORCH_DT[dtype])
    resg = q(rnd((B, Cin, Hh, W), 45), dtype)
    mask = q(rnd((B, Cin, Hh, W), 46), dtype)
    try:
        H.set_option("CONV_LC", 2)
        gx = ops.conv2d_dgrad(dtype, to_dev(gy, dtype), wt, None, (B, Hh, W, Cin), 3, 3, 1, 1)
        gx3 = ops.conv2d_dgrad(dtype, to_dev(gy, dtype), wt, to_dev(resg, dtype), (B, Hh, W, Cin), 3, 3, 1, 1, to_dev(mask, dtype))
    finally:
        H.set_option("CONV_LC", None)
    assert rel_err(from_dev(gx), xg.grad) < TOL[dtype]
    got3 = from_dev(gx3)
    assert rel_err(got3, (xg.grad + resg) * (mask > 0)) < TOL[dtype]
    assert float((got3 * (mask <= 0)).abs().max()) == 0.0


def test_conv_lc_is_the_automatic_choice_for_multi_round_launches():
    """The launch name recorded by the library's profiling layer tells which kernel ran: conv_lc for a launch of more than one
    round of workgroups, conv_rs for a single round (csrc/conv_lc.hip: dcf_conv3x3_lc_launch)."""
    ops, H = pkg("ops"), pkg("_hip")
    names = {}
    for B, tag in ((1, "single"), (8, "multi")):
        x = torch.zeros((B, 88, 100, 192), device="cuda", dtype=torch.bfloat16)
        w = torch.zeros((192, 3, 3, 192), device="cuda", dtype=torch.bfloat16)
        H.call("dcf_prof_reset")
        H.call("dcf_prof_enable", 1)
        try:
            ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, 192)
            torch.cuda.synchronize()
        finally:
            H.call("dcf_prof_enable", 0)
        names[tag] = [n for n in H.prof_read() if n.startswith("conv_fwd")]
        H.call("dcf_prof_reset")
    assert any("<rs" in n for n in names["single"]), names
    assert any("<lc" in n for n in names["multi"]), names
