"""fp8 (OCP e4m3) forward convolution path (BASELINE.json configs[4], cfg5): cast, weight images, MFMA kernel.

The reference has no fp8 path ("parity unpinned"); the check is a CPU statement of the kernel's own definition
(include/dcf_hip.h): operands rounded to e4m3 by torch's float8_e4m3fn conversion (round-to-nearest-even), products
and sums in fp64, scales applied after the sum."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import TORCH_DT, from_dev, pkg, rnd, to_dev

pytestmark = pytest.mark.gpu


def q8(t):
    """fp32 -> e4m3 -> fp32 on the CPU."""
    return t.to(torch.float8_e4m3fn).float()


def bits8(t):
    return t.to(torch.float8_e4m3fn).view(torch.uint8)


def act_scale_ref(amax):
    a = np.float32(amax)
    if not (a > 0 and np.isfinite(a)):
        return 1.0
    s = np.float32(224.0) / a
    u = np.frombuffer(np.float32(s).tobytes(), dtype=np.uint32)[0] & np.uint32(0x7F800000)
    return float(np.frombuffer(np.uint32(u).tobytes(), dtype=np.float32)[0])


def test_act_scale_rule():
    ops = pkg("ops")
    for amax in (0.0, 1e-30, 0.013, 0.5, 1.0, 3.7, 223.9, 224.0, 224.1, 448.0, 1e4, float("inf"), float("nan")):
        s = ops.fp8_act_scale(amax)
        assert s == act_scale_ref(amax)
        if 0 < amax < 1e30:
            assert amax * s <= 224.0 < amax * s * 2 and np.log2(s) == int(np.log2(s))


@pytest.mark.parametrize("dtype", [1, 2, 0])
def test_cast_fp8_bits_and_amax(dtype):
    ops = pkg("ops")
    x = rnd((3, 7, 9, 64), 21, -6.0, 6.0)
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-4, -3e-3, 447.0, -500.0, 0.0625, 17.0])     # zero, tiny, near / past the range
    xd = x.cuda().to(TORCH_DT[dtype])
    xq = xd.float().cpu()
    for amax_prev in (None, 500.0, 0.02):
        prev = None if amax_prev is None else torch.tensor([amax_prev], dtype=torch.float32, device="cuda")
        cur = torch.full((64,), 0.25, dtype=torch.float32, device="cuda")
        x8 = ops.cast_fp8(dtype, xd, prev, cur)
        s = act_scale_ref(amax_prev or 0.0)
        want = bits8((xq * s).clamp(-448.0, 448.0))
        assert torch.equal(x8.cpu(), want)
        assert float(cur.max().item()) == float(xq.abs().max()) and float(cur.min().item()) >= 0.25
    # a running maximum is kept, not overwritten
    cur = torch.full((64,), 1e6, dtype=torch.float32, device="cuda")
    ops.cast_fp8(dtype, xd, None, cur)
    assert float(cur.max().item()) == 1e6 and float(cur.min().item()) == 1e6


def conv_ref(xq, wq, wscale, sx, shift, res, stride, pad, relu):
    y = F.conv2d(xq.double(), wq.double(), None, stride, pad)
    y = y * (wscale.double().view(1, -1, 1, 1) / sx)
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    if res is not None:
        y = y + res.double()
    return (y.clamp_min(0) if relu else y).float()


CASES = [
    # B, H, W, Cin, Cout, k, stride, res, relu
    (2, 20, 24, 64, 64, 3, 1, True, True),
    (1, 17, 13, 64, 128, 3, 2, False, True),
    (2, 12, 40, 128, 64, 1, 1, False, False),
    (1, 33, 31, 192, 192, 3, 1, True, False),
    (1, 9, 11, 256, 96, 3, 1, False, True),        # Cout not a multiple of 64: 32-channel tiles
    (1, 16, 16, 64, 32, 1, 2, False, False),
    (3, 48, 64, 64, 128, 3, 1, False, True),       # enough pixels for the large tiles
    (1, 5, 6, 512, 256, 3, 1, True, True),
    # launches of >= 512 output tiles -- the only ones the product sends to this kernel (backend fp8_min_blocks)
    (1, 176, 200, 128, 128, 3, 1, True, True),     # LiDAR stage-3 body: 275 pixel tiles x 2 channel tiles
    (4, 88, 100, 192, 192, 3, 1, False, True),     # stage-4 body at batch 4: 275 x 3 tiles of 64 channels
    (4, 135, 240, 256, 128, 1, 1, False, False),   # cfg5 camera map (1080p / 8) at batch 4, 1x1: 1013 x 2 tiles
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("out_dtype", [1, 0])
def test_conv_fp8_matches_quantised_reference(case, out_dtype):
    B, Hh, W, Cin, Cout, k, stride, with_res, relu = case
    ops = pkg("ops")
    pad = k // 2
    x = rnd((B, Cin, Hh, W), 5, -2.0, 2.0)
    w = rnd((Cout, Cin, k, k), 6, -0.08, 0.08) * torch.linspace(0.2, 3.0, Cout).view(-1, 1, 1, 1)     # channel-dependent ranges
    shift = rnd((Cout,), 7, -0.5, 0.5)
    amax = float(x.abs().max()) * 1.3                       # "previous step" maximum: not this tensor's own
    sx = act_scale_ref(amax)
    wamax = w.abs().amax(dim=(1, 2, 3))
    sw = np.float32(448.0) / wamax
    wq = q8(w * sw.view(-1, 1, 1, 1))
    wscale = (wamax / np.float32(448.0)).float()
    xq = q8(x * sx)
    Ho, Wo = (Hh + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = rnd((B, Cout, Ho, Wo), 8, -1.0, 1.0) if with_res else None
    tdt = TORCH_DT[out_dtype]
    res_q = res.to(tdt).float() if with_res else None
    want = conv_ref(xq, wq, wscale, sx, shift, res_q, stride, pad, relu)
    # device: fp8 bits made on the host so that this test isolates the MFMA kernel
    x8 = bits8(x * sx).permute(0, 2, 3, 1).contiguous().cuda()
    w8 = bits8(w * sw.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous().cuda()
    xam = torch.tensor([amax], dtype=torch.float32, device="cuda")
    yam = torch.tensor([float(want.abs().max()) * 0.7], dtype=torch.float32, device="cuda")
    ycur = torch.zeros(64, device="cuda")
    y, y8 = ops.conv2d_fwd_fp8(out_dtype, x8, w8, wscale.cuda(), xam, shift.cuda(), to_dev(res, out_dtype) if with_res else None,
                               k, k, stride, pad, relu, Cout, want_y8=True, y8amax=yam, y8cur=ycur)
    got = from_dev(y)
    # the fused second output is the cast of y as stored, and its maximum is tracked
    assert torch.equal(y8, ops.cast_fp8(out_dtype, y, yam, None))
    assert float(ycur.max().item()) == float(y.float().abs().max().item())
    tol = 1e-5 if out_dtype == 0 else 8e-3
    err = float((got - want).abs().max() / want.abs().max())
    assert err < tol, err
    # the same through the device-side cast (bf16 activations in, scale from the device scalar)
    xd = to_dev(x, 1)
    x8b = ops.cast_fp8(1, xd, xam, None)
    assert torch.equal(x8b.cpu(), bits8(xd.float().cpu() * sx))


def test_fp8_path_close_to_bf16_conv():
    """End to end on one layer: quantisation error of the fp8 path against the bf16 kernel is at the e4m3 level."""
    ops = pkg("ops")
    B, Hh, W, Cin, Cout = 2, 40, 48, 128, 128
    x = rnd((B, Cin, Hh, W), 15, 0.0, 2.0)
    w = rnd((Cout, Cin, 3, 3), 16, -0.05, 0.05)
    xd = to_dev(x, 1)
    wd = w.permute(0, 2, 3, 1).contiguous().cuda().to(torch.bfloat16)
    y16 = from_dev(ops.conv2d_fwd(1, xd, wd, None, None, 3, 3, 1, 1, False, Cout))
    wamax = w.abs().amax(dim=(1, 2, 3))
    w8 = bits8(w * (448.0 / wamax).view(-1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous().cuda()
    cur = torch.zeros(64, device="cuda")
    ops.cast_fp8(1, xd, None, cur)                                   # step 0: collects the maximum
    prev = cur.max().reshape(1)
    x8 = ops.cast_fp8(1, xd, prev, None)                             # step 1: scaled by it
    y8 = from_dev(ops.conv2d_fwd_fp8(1, x8, w8, (wamax / 448.0).cuda(), prev, None, None, 3, 3, 1, 1, False, Cout))
    rel = float((y8 - y16).norm() / y16.norm())
    assert rel < 0.04, rel


def test_fp8_bad_arguments_fail_loudly():
    ops, Hm = pkg("ops"), pkg("_hip")
    x8 = torch.zeros((1, 8, 8, 32), dtype=torch.uint8, device="cuda")
    w8 = torch.zeros((64, 3, 3, 32), dtype=torch.uint8, device="cuda")
    ws = torch.ones(64, device="cuda")
    with pytest.raises(Hm.DcfError):
        ops.conv2d_fwd_fp8(1, x8, w8, ws, None, None, None, 3, 3, 1, 1, False, 64)       # Cin = 32: not a K=64 step
