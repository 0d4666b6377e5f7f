"""Operator layer: one Python function per C-ABI entry point of libdcf_hip.so.

Each function takes torch CUDA tensors (NHWC activations, compute dtype f32 or bf16),
allocates the output with torch (device memory = plumbing) and enqueues the HIP kernel on
torch's current stream.  No arithmetic happens in torch here.
"""
import ctypes

import numpy as np
import torch

from . import _hip as H


def _chk(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
        raise H.DcfError("%s must be a contiguous CUDA tensor" % name)
    return t


def conv_out_size(n, k, s, p):
    return (n + 2 * p - k) // s + 1


# ------------------------------------------------------------------ layout
def nchw_to_nhwc(x, dtype):
    B, C, Hh, W = x.shape
    _chk(x, "x")
    y = torch.empty((B, Hh, W, C), dtype=H.torch_dtype(dtype), device=x.device)
    H.call("dcf_nchw_to_nhwc", dtype, x, y, B, C, Hh, W, H.stream_ptr())
    return y


def nhwc_to_nchw(x, dtype):
    """NHWC tensor of the compute dtype -> NCHW fp32 (the reference's activation layout)."""
    B, Hh, W, C = x.shape
    _chk(x, "x")
    y = torch.empty((B, C, Hh, W), dtype=torch.float32, device=x.device)
    H.call("dcf_nhwc_to_nchw", dtype, x, y, B, C, Hh, W, H.stream_ptr())
    return y


def image_to_nhwc4(img_u8, dtype):
    B, C, Hh, W = img_u8.shape
    assert C == 3 and img_u8.dtype == torch.uint8
    _chk(img_u8, "image")
    y = torch.empty((B, Hh + 6, W + 8, 4), dtype=H.torch_dtype(dtype), device=img_u8.device)
    H.call("dcf_image_to_nhwc4", dtype, img_u8, y, B, Hh, W, H.stream_ptr())
    return y


# ------------------------------------------------------------------ convolution
def conv2d_fwd(dtype, x, w, shift, res, kh, kw, stride, pad, relu, cout):
    B, Hh, W, Cin = x.shape
    Ho, Wo = conv_out_size(Hh, kh, stride, pad), conv_out_size(W, kw, stride, pad)
    y = torch.empty((B, Ho, Wo, cout), dtype=x.dtype, device=x.device)
    H.call("dcf_conv2d_fwd", dtype, x, w, shift, res, y, B, Hh, W, Cin, Ho, Wo, cout, kh, kw, stride, pad, int(relu),
           H.stream_ptr())
    return y


def conv2d_fwd_rowscale(dtype, x, w, shift, rowscale, res, relu, cout):
    """1x1 conv2d_fwd whose shift is scaled per output pixel: y = act(conv + rowscale[m] * shift[c] + res); rowscale fp32 [B*H*W]."""
    B, Hh, W, Cin = x.shape
    if rowscale.numel() != B * Hh * W or rowscale.dtype != torch.float32:
        raise H.DcfError("conv2d_fwd_rowscale: rowscale must be fp32 with one entry per output pixel")
    y = torch.empty((B, Hh, W, cout), dtype=x.dtype, device=x.device)
    H.call("dcf_conv2d_fwd_rowscale", dtype, x, w, shift, _chk(rowscale, "rowscale"), res, y, B, Hh, W, Cin, Hh, W, cout, 1, 1, 1, 0, int(relu),
           H.stream_ptr())
    return y


def fp8_act_scale(amax):
    """The activation scale rule of the fp8 path (largest power of two s with amax*s <= 224)."""
    import ctypes
    out = ctypes.c_float(0.0)
    H.call("dcf_fp8_act_scale", float(amax), ctypes.addressof(out))
    return float(out.value)


def cast_fp8(dtype, x, amax_prev=None, amax_cur=None):
    """uint8 image (OCP e4m3 bits) of x * scale(amax_prev); amax_cur (device float[64]) collects partial maxima of |x|."""
    x8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    H.call("dcf_cast_fp8", dtype, x, x8, amax_prev, amax_cur, x.numel(), H.stream_ptr())
    return x8


def conv2d_fwd_fp8(out_dtype, x8, w8, wscale, xamax, shift, res, kh, kw, stride, pad, relu, cout, want_y8=False, y8amax=None, y8cur=None):
    """want_y8: also return the fp8 image of y (scale from y8amax, partial maxima into y8cur[64]) -> (y, y8)."""
    B, Hh, W, Cin = x8.shape
    Ho, Wo = conv_out_size(Hh, kh, stride, pad), conv_out_size(W, kw, stride, pad)
    y = torch.empty((B, Ho, Wo, cout), dtype=H.torch_dtype(out_dtype), device=x8.device)
    y8 = torch.empty((B, Ho, Wo, cout), dtype=torch.uint8, device=x8.device) if want_y8 else None
    H.call("dcf_conv2d_fwd_fp8", out_dtype, x8, w8, wscale, xamax, shift, res, y, y8, y8amax, y8cur, B, Hh, W, Cin, Ho, Wo, cout, kh, kw,
           stride, pad, int(relu), H.stream_ptr())
    return (y, y8) if want_y8 else y


def conv2d_dgrad(dtype, gy, wt, res, in_shape, kh, kw, stride, pad, mask=None):
    B, Hh, W, Cin = in_shape
    _, Ho, Wo, Cout = gy.shape
    gx = torch.empty((B, Hh, W, Cin), dtype=gy.dtype, device=gy.device)
    H.call("dcf_conv2d_dgrad", dtype, gy, wt, res, mask, gx, B, Hh, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, H.stream_ptr())
    return gx


def conv2d_dgrad_halfres(dtype, gy, wt, res, resq, in_shape, kh, kw, stride, pad, mask=None):
    """conv2d_dgrad of a stride-2 layer plus resq [B, ceil(H/2), ceil(W/2), Cin] added on the (2i, 2j) sub-grid of gx."""
    B, Hh, W, Cin = in_shape
    _, Ho, Wo, Cout = gy.shape
    if tuple(resq.shape) != (B, (Hh + 1) // 2, (W + 1) // 2, Cin):
        raise H.DcfError("conv2d_dgrad_halfres: resq %s does not sit on the even sub-grid of %s" % (tuple(resq.shape), tuple(in_shape)))
    gx = torch.empty((B, Hh, W, Cin), dtype=gy.dtype, device=gy.device)
    H.call("dcf_conv2d_dgrad_halfres", dtype, gy, wt, res, resq, mask, gx, B, Hh, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, H.stream_ptr())
    return gx


def conv3x3_chain_supported(dtype, B, Hh, W, C, nlayers):
    """Can dcf_conv3x3_chain run `nlayers` 3x3 / stride-1 layers of C channels on [B,Hh,W,C] in one launch?"""
    return bool(H.lib().dcf_conv3x3_chain_supported(dtype, B, Hh, W, C, nlayers))


def conv3x3_chain_workspace(dtype, B, Hh, W, C, nlayers, device):
    """Arrival counters of a chain launch, zeroed ONCE here (every launch leaves them zero); int32 tensor."""
    n = H.lib().dcf_conv3x3_chain_workspace_bytes(dtype, B, Hh, W, C, nlayers)
    if n == 0:
        raise H.DcfError("conv3x3_chain: unsupported shape %s x %d layers" % ((B, Hh, W, C), nlayers))
    return torch.zeros((n // 4,), dtype=torch.int32, device=device)


def conv3x3_chain(dtype, x, layers, flip, ws, table=None):
    """A chain of 3x3 / stride-1 / pad-1 layers with C channels in and out, each reading the one before it, in ONE launch
    (dcf_hip.h: dcf_conv3x3_chain).  x [B,H,W,C]; layers = list of (w, shift, res, mask, relu) where w / shift are tensors or
    raw device addresses and res / mask are tensors, None, or an int k = "the output of chain layer k" (k <= l - 2).
    flip = 1: input gradients (w = the [Cin][tap][Cout] images).  Returns the list of outputs (new tensors).
    table: a ctypes (H.ChainLayer * n) to reuse between calls (the caller keeps it alive)."""
    B, Hh, W, C = x.shape
    n = len(layers)
    outs = [torch.empty((B, Hh, W, C), dtype=x.dtype, device=x.device) for _ in range(n)]
    tab = table if table is not None else (H.ChainLayer * n)()
    cur = x.data_ptr()
    for l, (w, shift, res, mask, relu) in enumerate(layers):
        it = tab[l]
        it.x = cur
        it.w = H._ptr(w)
        it.shift = H._ptr(shift)
        it.res = outs[res].data_ptr() if type(res) is int else H._ptr(res)
        it.mask = outs[mask].data_ptr() if type(mask) is int else H._ptr(mask)
        cur = it.y = outs[l].data_ptr()
        it.relu = 1 if relu else 0
    import ctypes
    H.call("dcf_conv3x3_chain", dtype, ctypes.addressof(tab), n, B, Hh, W, C, int(flip), ws, H.stream_ptr())
    return outs


def conv3x3_chain_status(ws):
    """The give-up record of the last chain launches on workspace ws (0 = every wait was satisfied); synchronises."""
    return int(ws[1].item())


def conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, kh, kw, stride=1):
    return H.lib().dcf_conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, kh, kw, stride)


def conv2d_wgrad(dtype, x, gy, slabs, nsplit, kh, kw, stride, pad, gsum=None):
    B, Hh, W, Cin = x.shape
    _, Ho, Wo, Cout = gy.shape
    H.call("dcf_conv2d_wgrad", dtype, x, gy, slabs, gsum, nsplit, B, Hh, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, H.stream_ptr())
    return slabs


def stem7x7_fwd(dtype, img4, w, shift, relu, cout, Hh, W):
    B = img4.shape[0]
    Ho, Wo = conv_out_size(Hh, 7, 2, 3), conv_out_size(W, 7, 2, 3)
    y = torch.empty((B, Ho, Wo, cout), dtype=img4.dtype, device=img4.device)
    H.call("dcf_stem7x7_fwd", dtype, img4, w, shift, y, B, Hh, W, Ho, Wo, cout, int(relu), H.stream_ptr())
    return y


def stem7x7_wgrad(dtype, img4, gy, slabs, nsplit, Hh, W, gsum=None):
    B, Ho, Wo, Cout = gy.shape
    H.call("dcf_stem7x7_wgrad", dtype, img4, gy, slabs, gsum, nsplit, B, Hh, W, Ho, Wo, Cout, H.stream_ptr())
    return slabs


# ------------------------------------------------------------------ elementwise
def relu_bwd_chansum(dtype, gy, y, gsum, relu=True):
    C = gy.shape[-1]
    npix = gy.numel() // C
    H.call("dcf_relu_bwd_chansum", dtype, gy, y if relu else None, gsum, npix, C, int(relu), H.stream_ptr())
    return gy


def bn_workspace(C, device):
    return torch.empty((H.lib().dcf_bn_workspace_bytes(C),), dtype=torch.uint8, device=device)


def bn_train_fwd(dtype, x, gamma, beta, res, running_mean, running_var, relu, ws, eps=1e-5, momentum=0.1):
    """Returns (y, mean, invstd); running stats are updated in place when given."""
    C = x.shape[-1]
    npix = x.numel() // C
    y = torch.empty_like(x)
    mean = torch.empty((C,), dtype=torch.float32, device=x.device)
    invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
    H.call("dcf_bn_train_fwd", dtype, x, gamma, beta, res, y, mean, invstd, running_mean, running_var, npix, C, eps, momentum,
           int(relu), ws, H.stream_ptr())
    return y, mean, invstd


def bn_train_bwd(dtype, g, x, mean, invstd, gamma, dgamma, dbeta, ws):
    C = x.shape[-1]
    dx = torch.empty_like(x)
    H.call("dcf_bn_train_bwd", dtype, g, x, mean, invstd, gamma, dgamma, dbeta, dx, x.numel() // C, C, ws, H.stream_ptr())
    return dx


def resize_bilinear_fwd(dtype, x, out_hw, align_corners, add=None):
    B, Hi, Wi, C = x.shape
    Ho, Wo = out_hw
    y = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
    H.call("dcf_resize_bilinear_fwd", dtype, x, add, y, B, Hi, Wi, Ho, Wo, C, int(align_corners), H.stream_ptr())
    return y


def resize_bilinear_bwd(dtype, gy, in_hw, align_corners):
    B, Ho, Wo, C = gy.shape
    Hi, Wi = in_hw
    gx = torch.empty((B, Hi, Wi, C), dtype=gy.dtype, device=gy.device)
    H.call("dcf_resize_bilinear_bwd", dtype, gy, gx, B, Hi, Wi, Ho, Wo, C, int(align_corners), H.stream_ptr())
    return gx


def maxpool_fwd(dtype, x):
    B, Hh, W, C = x.shape
    Ho, Wo = (Hh - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
    H.call("dcf_maxpool3x3s2_fwd", dtype, x, y, B, Hh, W, Ho, Wo, C, H.stream_ptr())
    return y


def maxpool_fwd_idx(dtype, x):
    """(y, idx): pooled output and the arg-max positions (uint32 [B,Ho,Wo,C/4], one byte per channel)."""
    B, Hh, W, C = x.shape
    Ho, Wo = (Hh - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
    idx = torch.empty((B, Ho, Wo, C // 4), dtype=torch.int32, device=x.device)
    H.call("dcf_maxpool3x3s2_fwd_idx", dtype, x, y, idx, B, Hh, W, Ho, Wo, C, H.stream_ptr())
    return y, idx


def maxpool_bwd_idx(dtype, idx, gy, in_shape):
    B, Hh, W, C = in_shape
    _, Ho, Wo, _ = gy.shape
    gx = torch.empty(in_shape, dtype=gy.dtype, device=gy.device)
    H.call("dcf_maxpool3x3s2_bwd_idx", dtype, idx, gy, gx, B, Hh, W, Ho, Wo, C, H.stream_ptr())
    return gx


def maxpool_bwd(dtype, x, y, gy):
    B, Hh, W, C = x.shape
    _, Ho, Wo, _ = gy.shape
    gx = torch.empty_like(x)
    H.call("dcf_maxpool3x3s2_bwd", dtype, x, y, gy, gx, B, Hh, W, Ho, Wo, C, H.stream_ptr())
    return gx


def head_fwd(dtype, head, anchors):
    B, h, w, Cp = head.shape
    pred = torch.empty((B, 32, h, w), dtype=torch.float32, device=head.device)
    H.call("dcf_head_fwd", dtype, head, Cp, anchors, pred, B, h, w, H.stream_ptr())
    return pred


def head_bwd(dtype, head, anchors, pred, gpred):
    B, h, w, Cp = head.shape
    ghead = torch.empty_like(head)
    H.call("dcf_head_bwd", dtype, head, Cp, anchors, pred, _chk(gpred, "gpred"), ghead, B, h, w, H.stream_ptr())
    return ghead


def cast(src, dtype_dst):
    dst = torch.empty(src.shape, dtype=H.torch_dtype(dtype_dst), device=src.device)
    H.call("dcf_cast", H.dtype_code(src.dtype), src, dtype_dst, dst, src.numel(), H.stream_ptr())
    return dst


def adam_step(params, grads, m, v, lr, beta1, beta2, eps, step, gscale=1.0):
    H.call("dcf_adam_step", params, grads, m, v, params.numel(), lr, beta1, beta2, eps, step, gscale, H.stream_ptr())


# ------------------------------------------------------------------ geometry
class GridSpec(object):
    """Grid constants of CarlaDataset.__init__ (data_import_carla.py:35-43) and the filter
    thresholds of :215-226 -- host-side integer arithmetic only."""

    def __init__(self, cfg):
        L, W, Cz = cfg["voxel_length"], cfg["voxel_width"], cfg["voxel_channel"]
        self.dims = (Cz, L, W)
        self.xs = int(L / (cfg["lidar_x_max"] - cfg["lidar_x_min"]))
        self.ys = int(W / (cfg["lidar_y_max"] - cfg["lidar_y_min"]))
        self.zs = int(Cz / (cfg["lidar_z_max"] - cfg["lidar_z_min"]))
        self.xo = int(-cfg["lidar_x_min"] * self.xs)
        self.yo = int(-cfg["lidar_y_min"] * self.ys)
        self.zo = int(-cfg["lidar_z_min"] * self.zs)
        d = cfg["delta"]
        self.lim = np.array([cfg["lidar_x_min"], cfg["lidar_x_max"] - d, cfg["lidar_y_min"], cfg["lidar_y_max"] - d,
                             cfg["lidar_z_min"], cfg["lidar_z_max"] - d], dtype=np.float64).astype(np.float32)
        self.aff = np.array([self.xs, self.xo, self.ys, self.yo, self.zs, self.zo], dtype=np.float32)
        self.image_height = cfg["image_height"]
        self.image_width = cfg["image_width"]


def range_filter(pts, lim):
    n = pts.shape[0]
    dev = pts.device
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev)
    src = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((H.lib().dcf_compact_workspace_bytes(n),), dtype=torch.uint8, device=dev)
    H.call("dcf_range_filter", _chk(pts, "pts"), n, H.host_f32(lim), out, src, cnt, ws, H.stream_ptr())
    return out, src, cnt


def voxelize(pts, lim, aff, dims, mode=H.VOXEL_COMPAT, owner_ws=None, out=None):
    Cz, L, W = dims
    dev = pts.device
    grid = torch.empty((Cz, L, W), dtype=torch.float32, device=dev) if out is None else _chk(out, "out")
    if mode in (H.VOXEL_COMPAT, H.VOXEL_COMPAT_ROUNDS) and owner_ws is None:
        owner_ws = torch.zeros((2, Cz * L * W), dtype=torch.int32, device=dev)
    H.call("dcf_voxelize", _chk(pts, "pts"), pts.shape[0], H.host_f32(lim), H.host_f32(aff), Cz, L, W, mode, grid, owner_ws,
           H.stream_ptr())
    return grid


def voxelize_batch(pts_list, lim, aff, dims, owner_ws, out):
    """Compat-mode grids of B frames (claim / gather / release, one launch each for all frames): out [B,Cz,L,W] fp32,
    owner_ws int32 [B,2,Cz*L*W] (zero on entry and on exit)."""
    Cz, L, W = dims
    B = len(pts_list)
    ptrs = (ctypes.c_void_p * B)(*[_chk(p, "pts").data_ptr() for p in pts_list])
    ns = (ctypes.c_int * B)(*[p.shape[0] for p in pts_list])
    H.call("dcf_voxelize_batch", ctypes.addressof(ptrs), ctypes.addressof(ns), B, H.host_f32(lim), H.host_f32(aff), Cz, L, W, _chk(out, "out"),
           owner_ws, H.stream_ptr())
    return out


def voxelize_batch_nhwc(dtype, pts_list, lim, aff, dims, owner_ws, out):
    """The same grids as the engine's input image: out [B,L,W,Cz] in the compute dtype (= nchw_to_nhwc of voxelize_batch)."""
    Cz, L, W = dims
    B = len(pts_list)
    ptrs = (ctypes.c_void_p * B)(*[_chk(p, "pts").data_ptr() for p in pts_list])
    ns = (ctypes.c_int * B)(*[p.shape[0] for p in pts_list])
    H.call("dcf_voxelize_batch_nhwc", dtype, ctypes.addressof(ptrs), ctypes.addressof(ns), B, H.host_f32(lim), H.host_f32(aff), Cz, L, W,
           _chk(out, "out"), owner_ws, H.stream_ptr())
    return out


def project_filter(pts, lim, crt, ulim, vlim, mode=H.PROJ_COMPAT, n_out=None, want_src=False, out=None):
    """Returns (uv [n_out,2], xyz [n_out,3], count int32[1] (device), src or None); rows past count are zero.
    out: optional (uv, xyz, count) to write into -- zero-filled, contiguous, at least as many rows as input points (a
    batch's frames then land side by side in one tensor without a stack copy)."""
    n = pts.shape[0]
    dev = pts.device
    n_out = n if n_out is None else n_out
    if n_out < n:
        raise H.DcfError("project_filter: output rows (%d) < input points (%d)" % (n_out, n))
    if out is not None:
        uv, xyz, cnt = out
        if uv.shape[0] < n or xyz.shape[0] < n or not (uv.is_contiguous() and xyz.is_contiguous()):
            raise H.DcfError("project_filter: out tensors must be contiguous with at least %d rows" % n)
    else:
        uv = torch.zeros((max(n_out, 1), 2), dtype=torch.float32, device=dev)
        xyz = torch.zeros((max(n_out, 1), 3), dtype=torch.float32, device=dev)
        cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    src = torch.empty((max(n, 1),), dtype=torch.int32, device=dev) if want_src else None
    ws = torch.empty((H.lib().dcf_compact_workspace_bytes(n),), dtype=torch.uint8, device=dev)
    H.call("dcf_project_filter", _chk(pts, "pts"), n, H.host_f32(lim), H.host_f32(crt).reshape(-1), float(ulim), float(vlim), mode,
           uv, xyz, src, cnt, ws, H.stream_ptr())
    return uv, xyz, cnt, src


def project_filter_batch(pts_list, lim, crts, ulim, vlim, mode, uv_all, xyz_all, cnt_all, ws=None):
    """dcf_project_filter for the (<= 8) frames of a batch in one launch per phase: pts_list = B device tensors [n_b,3]; crts [B,4,3]
    (or [B,12]) host floats, one matrix per frame; uv_all [B,rows,2], xyz_all [B,rows,3] (zero-filled by the caller: rows past a
    frame's count are left untouched), cnt_all int32 [B].  Same values as one project_filter per frame."""
    import ctypes
    B = len(pts_list)
    rows = xyz_all.shape[1]
    ns = [int(p.shape[0]) for p in pts_list]
    if max(ns) > rows or uv_all.shape[1] != rows or not (uv_all.is_contiguous() and xyz_all.is_contiguous() and cnt_all.is_contiguous()):
        raise H.DcfError("project_filter_batch: contiguous outputs with at least %d rows per frame" % max(ns))
    ptrs = (ctypes.c_void_p * B)(*[_chk(p, "pts").data_ptr() if p.shape[0] else None for p in pts_list])
    cnts = (ctypes.c_int32 * B)(*ns)
    crt = np.ascontiguousarray(np.asarray(crts, dtype=np.float32).reshape(B, 12))
    need = B * H.lib().dcf_compact_workspace_bytes(max(max(ns), 1))
    if ws is None or ws.numel() < need:
        ws = torch.empty((need,), dtype=torch.uint8, device=xyz_all.device)
    H.call("dcf_project_filter_batch", ctypes.addressof(ptrs), ctypes.addressof(cnts), B, H.host_f32(lim), crt, float(ulim), float(vlim), int(mode),
           uv_all, xyz_all, rows, cnt_all, ws, H.stream_ptr())
    return ws


def knn_bev(xyz, cnt, K, h, w, stride, aff, rmax=None, ws=None, out=None):
    """out: optional int32 [K,h,w] tensor to write into (e.g. a frame's slice of a batch tensor)."""
    n_max = xyz.shape[0]
    dev = xyz.device
    idx = torch.empty((K, h, w), dtype=torch.int32, device=dev) if out is None else _chk(out, "out")
    if ws is None:
        ws = torch.empty((H.lib().dcf_knn_workspace_bytes(n_max, h, w),), dtype=torch.uint8, device=dev)
    r2 = -1.0 if rmax is None else float(np.float32(rmax) * np.float32(rmax))
    H.call("dcf_knn_bev", _chk(xyz, "xyz"), cnt, n_max, K, h, w, stride, float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]),
           r2, idx, ws, H.stream_ptr())
    return idx


def knn_ws_stride(n_max, h, w):
    """Bytes between the per-frame workspaces of knn_bev_batch."""
    return (H.lib().dcf_knn_workspace_bytes(n_max, h, w) + 255) // 256 * 256


def knn_bev_batch(xyz, cnt, K, h, w, stride, aff, rmax=None, ws=None, out=None):
    """All frames of a batch in one launch per phase: xyz [B,n_max,3], cnt int32 [B] -> idx int32 [B,K,h,w]
    (= knn_bev frame by frame).  ws: uint8 [B, knn_ws_stride(...)]."""
    B, n_max = xyz.shape[0], xyz.shape[1]
    dev = xyz.device
    st = knn_ws_stride(n_max, h, w)
    idx = torch.empty((B, K, h, w), dtype=torch.int32, device=dev) if out is None else _chk(out, "out")
    if ws is None:
        ws = torch.empty((B, st), dtype=torch.uint8, device=dev)
    if ws.shape[0] < B or ws.stride(0) != st:
        raise H.DcfError("knn_bev_batch: workspace must be [B, %d] bytes" % st)
    r2 = -1.0 if rmax is None else float(np.float32(rmax) * np.float32(rmax))
    H.call("dcf_knn_bev_batch", _chk(xyz, "xyz"), _chk(cnt, "cnt"), B, n_max, K, h, w, stride, float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]),
           r2, idx, ws, st, H.stream_ptr())
    return idx


def knn_bev_batch_shared(xyz, cnt, K, h, w, stride, fine, ws_fine, aff, rmax=None, ws=None, out=None):
    """A coarser site of the same batch: own cell sort (ws) + a search that serves dense regions from the cells of a finer site:
    fine = (h, w, stride) of the knn_bev_batch call whose workspace ws_fine ([B, knn_ws_stride(n_max, fine_h, fine_w)] bytes,
    untouched since) is read.  Same indices as knn_bev_batch."""
    B, n_max = xyz.shape[0], xyz.shape[1]
    fh, fw, fs = fine
    stf = knn_ws_stride(n_max, fh, fw)
    if ws_fine.shape[0] < B or ws_fine.stride(0) != stf:
        raise H.DcfError("knn_bev_batch_shared: the fine site's workspace must be [B, %d] bytes" % stf)
    st = knn_ws_stride(n_max, h, w)
    if ws is None:
        ws = torch.empty((B, st), dtype=torch.uint8, device=xyz.device)
    if ws.shape[0] < B or ws.stride(0) != st:
        raise H.DcfError("knn_bev_batch_shared: workspace must be [B, %d] bytes" % st)
    idx = torch.empty((B, K, h, w), dtype=torch.int32, device=xyz.device) if out is None else _chk(out, "out")
    r2 = -1.0 if rmax is None else float(np.float32(rmax) * np.float32(rmax))
    H.call("dcf_knn_bev_batch_shared", _chk(xyz, "xyz"), _chk(cnt, "cnt"), B, n_max, K, h, w, stride, fh, fw, fs, float(aff[0]), float(aff[1]),
           float(aff[2]), float(aff[3]), r2, idx, ws, st, ws_fine, stf, H.stream_ptr())
    return idx


def knn_bev_sites(xyz, cnt, K, sites, aff, rmax=None):
    """Every fusion site of a batch in one call (dcf_knn_bev_sites): the cell sort of all sites in one launch per phase, then
    the searches.  sites = list of (h, w, stride, fine, ws, out): fine = index of an earlier, finer site whose cells serve this
    site's dense pixels (as knn_bev_batch_shared) or -1 (as knn_bev_batch); ws uint8 [B, knn_ws_stride(n_max, h, w)]; out int32
    [B, K, h, w].  Same maps as the per-site calls."""
    B, n_max = xyz.shape[0], xyz.shape[1]
    arr = (H.KnnSite * len(sites))()
    for i, (h, w, stride, fine, ws, out) in enumerate(sites):
        st = knn_ws_stride(n_max, h, w)
        if ws.shape[0] < B or ws.stride(0) != st or ws.dtype != torch.uint8:
            raise H.DcfError("knn_bev_sites: site %d: workspace must be uint8 [B, %d]" % (i, st))
        if tuple(out.shape) != (B, K, h, w) or out.dtype != torch.int32 or not out.is_contiguous():
            raise H.DcfError("knn_bev_sites: site %d: out must be a contiguous int32 [B, K, h, w]" % i)
        arr[i] = H.KnnSite(h, w, stride, fine, out.data_ptr(), ws.data_ptr(), st)
    r2 = -1.0 if rmax is None else float(np.float32(rmax) * np.float32(rmax))
    H.call("dcf_knn_bev_sites", _chk(xyz, "xyz"), _chk(cnt, "cnt"), B, n_max, K, ctypes.addressof(arr), len(sites), float(aff[0]), float(aff[1]),
           float(aff[2]), float(aff[3]), r2, H.stream_ptr())
    return [t[5] for t in sites]


# ------------------------------------------------------------------ fusion
def point_sample_fwd(dtype, fmap, uv, cnt, n_max, out=None):
    """out: optional [n_max, Cf] tensor to write into (a frame's slice of a batch tensor); every row is written (zeros past cnt)."""
    Hf, Wf, Cf = fmap.shape
    fp = (torch.empty if n_max > 0 else torch.zeros)((max(n_max, 1), Cf), dtype=fmap.dtype, device=fmap.device) if out is None else _chk(out, "out")
    H.call("dcf_point_sample_fwd", dtype, fmap, Hf, Wf, Cf, uv, cnt, n_max, fp, H.stream_ptr())
    return fp


def point_sample_bwd(dtype, gfp, uv, cnt, n_max, gfmap):
    Hf, Wf, Cf = gfmap.shape
    H.call("dcf_point_sample_bwd", dtype, gfp, Hf, Wf, Cf, uv, cnt, n_max, gfmap, H.stream_ptr())
    return gfmap


def point_sample_fwd_batch(dtype, fmap, uv, cnt, n_max, out):
    """All frames in one launch: fmap [B,Hf,Wf,Cf], uv [B,rows,2], cnt int32 [B], out [B,n_max,Cf] (every row is written: zeros past a frame's count)."""
    B, Hf, Wf, Cf = fmap.shape
    H.call("dcf_point_sample_fwd_batch", dtype, _chk(fmap, "fmap"), Hf, Wf, Cf, _chk(uv, "uv"), uv.stride(0), cnt, n_max, _chk(out, "out"), B, H.stream_ptr())
    return out


def point_sample_bwd_batch(dtype, gfp, uv, cnt, n_max, gfmap):
    """gfp [B,n_max,Cf], gfmap fp32 [B,Hf,Wf,Cf] (accumulated into)."""
    B, Hf, Wf, Cf = gfmap.shape
    H.call("dcf_point_sample_bwd_batch", dtype, _chk(gfp, "gfp"), Hf, Wf, Cf, _chk(uv, "uv"), uv.stride(0), cnt, n_max, _chk(gfmap, "gfmap"), B, H.stream_ptr())
    return gfmap


def fusion_gather_fwd_batch(dtype, P, xyz, idx, stride, aff, w1d, b1, hsum, cnt):
    """P [B,rows,Cb], xyz [B,n,3], idx int32 [B,K,h,w] -> hsum [B,h,w,Cb], cnt fp32 [B,h*w] (written)."""
    B, K, h, w = idx.shape
    Cb = P.shape[2]
    H.call("dcf_fusion_gather_fwd_batch", dtype, _chk(P, "P"), P.shape[1], _chk(xyz, "xyz"), xyz.stride(0), _chk(idx, "idx"), K, h, w, stride,
           float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]), w1d, b1, Cb, _chk(hsum, "hsum"), _chk(cnt, "cnt"), B, H.stream_ptr())
    return hsum, cnt


def fusion_gather_bwd_inv_batch(dtype, P, xyz, inv, n_max, g0, khw, stride, aff, w1d, b1, ghsum, gP, gw1d, gb1, ws=None):
    """fusion_gather_bwd_inv for the B frames of a batch in one launch: their maps are g0 .. g0 + B - 1 of the fusion_invert call;
    P [B,rows,Cb], xyz [B,n,3], ghsum [B,h,w,Cb], gP fp32 [B,rows,Cb]."""
    start, ent = inv
    B, rows, Cb = P.shape
    seg = start[g0 * (n_max + 1):]
    H.call("dcf_fusion_gather_bwd_inv_batch", dtype, _chk(P, "P"), rows, _chk(xyz, "xyz"), xyz.stride(0), seg, seg[n_max:], n_max + 1, ent[0], ent[1],
           khw[0] * khw[1] * khw[2], khw[1], khw[2], stride, float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]), w1d, b1, Cb,
           _chk(ghsum, "ghsum"), _chk(gP, "gP"), gw1d, gb1, ws, B, H.stream_ptr())


_DWS_BYTES = {}


def fusion_bwd_direct_workspace(device, max_entries, Cb, B):
    """Zeroed workspace of fusion_gather_bwd_direct_batch for maps of up to max_entries (= K*h*w) pairs (left zero by the kernel)."""
    return torch.zeros((max(H.lib().dcf_fusion_gather_bwd_direct_workspace_bytes(max_entries, Cb, B) // 4, 1),), dtype=torch.float32, device=device)


def fusion_gather_bwd_direct_batch(dtype, P, xyz, inv, n_max, g0, khw, stride, aff, w1d, b1, ghsum, gP, gw1d, gb1, ws, dws):
    """fusion_gather_bwd_inv_batch with gP [B,rows,Cb] in the compute dtype, ZERO on entry: rows are stored whole by their single
    writer; the few rows whose pairs cross a slice boundary go through dws (fusion_bwd_direct_workspace)."""
    start, ent = inv
    B, rows, Cb = P.shape
    me = khw[0] * khw[1] * khw[2]
    if gP.dtype != P.dtype or gP.shape != P.shape:
        raise H.DcfError("fusion_gather_bwd_direct_batch: gP must look like P")
    need = _DWS_BYTES.get((me, Cb, B))
    if need is None:
        need = _DWS_BYTES[(me, Cb, B)] = H.lib().dcf_fusion_gather_bwd_direct_workspace_bytes(me, Cb, B)
    if dws.numel() * 4 < need:
        raise H.DcfError("fusion_gather_bwd_direct_batch: workspace too small")
    # (raw addresses of the map's start segment and of the two pair arrays: four tensor views per call otherwise)
    seg = start.data_ptr() + 4 * g0 * (n_max + 1)
    H.call("dcf_fusion_gather_bwd_direct_batch", dtype, _chk(P, "P"), rows, _chk(xyz, "xyz"), xyz.stride(0), seg, seg + 4 * n_max, n_max + 1,
           ent.data_ptr(), ent.data_ptr() + 4 * ent.stride(0), me, khw[1], khw[2], stride, float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]),
           w1d, b1, Cb, _chk(ghsum, "ghsum"), _chk(gP, "gP"), gw1d, gb1, ws, dws, B, H.stream_ptr())


def fusion_gather_fwd(dtype, P, xyz, idx, stride, aff, w1d, b1, out=None):
    """out: optional (hsum [h,w,Cb], cnt [h*w]) to write into (a frame's slices of batch tensors)."""
    K, h, w = idx.shape
    Cb = P.shape[1]
    if out is not None:
        hsum, cnt = _chk(out[0], "hsum"), _chk(out[1], "cnt")
    else:
        hsum = torch.empty((h, w, Cb), dtype=P.dtype, device=P.device)
        cnt = torch.empty((h * w,), dtype=torch.float32, device=P.device)
    H.call("dcf_fusion_gather_fwd", dtype, P, xyz, idx, K, h, w, stride, float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]),
           w1d, b1, Cb, hsum, cnt, H.stream_ptr())
    return hsum, cnt


def fusion_gather_bwd(dtype, P, xyz, idx, stride, aff, w1d, b1, ghsum, gP, gw1d, gb1):
    K, h, w = idx.shape
    Cb = P.shape[1]
    H.call("dcf_fusion_gather_bwd", dtype, P, xyz, idx, K, h, w, stride, float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]),
           w1d, b1, Cb, ghsum, gP, gw1d, gb1, H.stream_ptr())


def fusion_invert_sizes(maps, n_max):
    """(elements of start, pairs, workspace bytes) of fusion_invert for these maps."""
    return len(maps) * (n_max + 1), sum(t.numel() for t in maps), H.lib().dcf_fusion_invert_workspace_bytes(n_max, len(maps))


def fusion_invert(maps, n_max, out=None):
    """maps: list of KNN maps [K,h,w] (sites x frames of a step).  Returns (start [len(maps)*(n_max+1)], ent [2, pairs]):
    the (pixel, point) pairs of all maps sorted by (map, point) -- see dcf_fusion_invert.
    out: optional (start, ent, ws) buffers of fusion_invert_sizes() to write into."""
    K = maps[0].shape[0]
    dev = maps[0].device
    tab = (H.KnnMap * len(maps))()
    total = 0
    for i, t in enumerate(maps):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.shape[0] == K
        tab[i] = H.KnnMap(t.data_ptr(), t.shape[1], t.shape[2])
        total += t.numel()
    if out is not None:
        start, ent, ws = out
    else:
        start = torch.empty((len(maps) * (n_max + 1),), dtype=torch.int32, device=dev)
        ent = torch.empty((2, total), dtype=torch.int32, device=dev)
        ws = torch.empty((H.lib().dcf_fusion_invert_workspace_bytes(n_max, len(maps)),), dtype=torch.uint8, device=dev)
    H.call("dcf_fusion_invert", ctypes.addressof(tab), len(maps), K, n_max, start, ent[0], ent[1], ws, H.stream_ptr())
    return start, ent


def fusion_bwd_workspace(device, Cb=256):
    """Zeroed workspace of fusion_gather_bwd_inv's slotted dW1d / db1 reduction (left zero: reusable by calls ordered on one stream)."""
    return torch.zeros((H.lib().dcf_fusion_gather_bwd_workspace_bytes(Cb) // 4,), dtype=torch.float32, device=device)


def fusion_gather_bwd_inv(dtype, P, xyz, inv, n_max, g, khw, stride, aff, w1d, b1, ghsum, gP, gw1d, gb1, ws=None):
    """n_max: the point-id range fusion_invert was called with; g: index of this (site, frame) map in that call;
    khw = (K, h, w) of the map.  P / gP may hold fewer rows than n_max (only ids below the valid count occur).
    ws: fusion_bwd_workspace() -> gw1d / gb1 through 16 accumulator copies folded by the last workgroup (16 instead of 256
    same-address atomics per word)."""
    start, ent = inv
    Cb = P.shape[1]
    seg = start[g * (n_max + 1):]
    if ws is not None and ws.numel() * 4 < H.lib().dcf_fusion_gather_bwd_workspace_bytes(Cb):
        raise H.DcfError("fusion_gather_bwd_inv: workspace too small for %d channels" % Cb)
    H.call("dcf_fusion_gather_bwd_inv", dtype, P, xyz, seg, seg[n_max:], ent[0], ent[1], khw[0] * khw[1] * khw[2], khw[1], khw[2], stride,
           float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]), w1d, b1, Cb, ghsum, gP, gw1d, gb1, ws, H.stream_ptr())


def rowscale_bias_fwd(dtype, y, cnt, b2):
    C = y.shape[-1]
    H.call("dcf_rowscale_bias_fwd", dtype, y, cnt, b2, y.numel() // C, C, H.stream_ptr())
    return y


def rowscale_bias_bwd(dtype, gy, cnt, gb2):
    C = gy.shape[-1]
    H.call("dcf_rowscale_bias_bwd", dtype, gy, cnt, gb2, gy.numel() // C, C, H.stream_ptr())


def relu_mask_rowscale_bwd(dtype, gy, y, cnt, gb2):
    """gout = gy * (y > 0) as a NEW tensor; gb2[c] += sum_p cnt[p] * gy[p][c] (one pass over gy: dcf_relu_mask_rowscale_bwd)."""
    C = gy.shape[-1]
    gout = torch.empty_like(gy)
    H.call("dcf_relu_mask_rowscale_bwd", dtype, _chk(gy, "gy"), _chk(y, "y"), cnt, gout, gb2, gy.numel() // C, C, H.stream_ptr())
    return gout


def fusion_gather_bwd_pts(dtype, P, xyz, inv, n_max, g, khw, stride, aff, w1d, b1, ghsum, gP, gw1d, gb1):
    """fusion_gather_bwd_inv with one writer per point row: gP [n_rows, Cb] in the compute dtype, fully written (no zero-fill
    before, no cast after).  Arguments as fusion_gather_bwd_inv."""
    start, ent = inv
    Cb = P.shape[1]
    seg = start[g * (n_max + 1):]
    H.call("dcf_fusion_gather_bwd_pts", dtype, P, xyz, seg, P.shape[0], ent[0], ent[1], khw[0] * khw[1] * khw[2], khw[1], khw[2], stride,
           float(aff[0]), float(aff[1]), float(aff[2]), float(aff[3]), w1d, b1, Cb, ghsum, gP, gw1d, gb1, H.stream_ptr())


# ------------------------------------------------------------------ evaluation post-processing
EVAL_NMS_CAP = 4096


def eval_score_filter(pred, threshold, cap=EVAL_NMS_CAP):
    """test.py:88-108 on the device: pred [B,32,h,w] fp32 -> (boxes [B,cap,7], count int32 [B]); per sample anchor 0's boxes
    with score > threshold in raster order, then anchor 1's."""
    B, C, h, w = pred.shape
    if C != 32 or pred.dtype != torch.float32:
        raise H.DcfError("eval_score_filter: pred must be the model output [B,32,h,w] fp32")
    pred = _chk(pred.contiguous(), "pred")
    boxes = torch.zeros((B, cap, 7), dtype=torch.float32, device=pred.device)
    count = torch.zeros((B,), dtype=torch.int32, device=pred.device)
    H.call("dcf_eval_score_filter", pred, B, h, w, float(threshold), cap, boxes, count, H.stream_ptr())
    return boxes, count


def eval_nms(boxes, mode, iou_threshold=0.01, count=None):
    """Greedy suppression in input order (test.py:110-175): boxes [n,7] fp32 on the device -> keep flags int32 [n].
    mode "sat" | "iou"; count: optional int32 [1] device tensor with the number of valid rows."""
    n = boxes.shape[0]
    if n > EVAL_NMS_CAP:
        raise H.DcfError("eval_nms: at most %d boxes per call (got %d)" % (EVAL_NMS_CAP, n))
    boxes = _chk(boxes.float().contiguous(), "boxes")
    keep = torch.zeros((max(n, 1),), dtype=torch.int32, device=boxes.device)
    nkeep = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
    ws = torch.empty((H.lib().dcf_eval_nms_workspace_bytes(max(n, 1)),), dtype=torch.uint8, device=boxes.device)
    H.call("dcf_eval_nms", boxes, count, n, {"sat": 0, "iou": 1}[mode], float(iou_threshold), keep, nkeep, ws, H.stream_ptr())
    return keep[:n], nkeep


def eval_match(pred_boxes, ref_boxes, thresholds, tp):
    """tp[t] += number of pred_boxes [n,7] whose bird's-eye IoU with a labelled row of ref_boxes [R,9] exceeds thresholds[t]
    (fp64 device tensor); tp int32 [len(thresholds)] on the device (test.py:177-206)."""
    n = pred_boxes.shape[0]
    if n == 0:
        return tp
    H.call("dcf_eval_match", _chk(pred_boxes.float().contiguous(), "pred_boxes"), n, _chk(ref_boxes.float().contiguous(), "ref_boxes"),
           ref_boxes.shape[0], thresholds, thresholds.shape[0], tp, H.stream_ptr())
    return tp
