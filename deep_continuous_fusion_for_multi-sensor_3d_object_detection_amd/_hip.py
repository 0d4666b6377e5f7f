"""ctypes binding of libdcf_hip.so (include/dcf_hip.h) -- the only compute path.

There is no CPU or eager-torch fallback: if the shared library is missing or a call
fails, an exception is raised.  torch is used for device memory and streams only
(`tensor.data_ptr()`, `torch.cuda.current_stream()`).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DCF_HIP_LIB: another build of the same library, e.g. an A/B variant of one kernel file linked beside the product's)
LIB_PATH = os.environ.get("DCF_HIP_LIB") or os.path.join(_HERE, "libdcf_hip.so")
CSRC = os.path.join(_HERE, "csrc")

F32, BF16, F16 = 0, 1, 2
VOXEL_COMPAT, VOXEL_ACCUM, VOXEL_COMPAT_ROUNDS, VOXEL_OCCUPANCY = 0, 1, 2, 3
PROJ_COMPAT, PROJ_CORRECT = 0, 1

c_int, c_float, c_i64, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_size_t, ctypes.c_void_p
P = c_void_p

# name -> (restype, argtypes); mirrors include/dcf_hip.h one to one
SIGNATURES = {
    "dcf_last_error": (ctypes.c_char_p, []),
    "dcf_version": (c_int, []),
    "dcf_set_option": (c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "dcf_prof_enable": (c_int, [c_int]),
    "dcf_prof_reset": (c_int, []),
    "dcf_prof_calibrate": (c_int, [P, c_int]),
    "dcf_prof_read": (c_int, [P, P, P, P, c_int]),
    "dcf_prof_read2": (c_int, [P, P, P, P, P, c_int]),
    "dcf_compact_workspace_bytes": (c_size_t, [c_int]),
    "dcf_range_filter": (c_int, [P, c_int, P, P, P, P, P, P]),
    "dcf_voxelize_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dcf_voxelize": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "dcf_voxelize_batch": (c_int, [P, P, c_int, P, P, c_int, c_int, c_int, P, P, P]),
    "dcf_voxelize_batch_nhwc": (c_int, [c_int, P, P, c_int, P, P, c_int, c_int, c_int, P, P, P]),
    "dcf_project_filter": (c_int, [P, c_int, P, P, c_float, c_float, c_int, P, P, P, P, P, P]),
    "dcf_project_filter_batch": (c_int, [P, P, c_int, P, P, c_float, c_float, c_int, P, P, c_int, P, P, P]),
    "dcf_knn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dcf_knn_bev": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, P, P, P]),
    "dcf_knn_bev_batch": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, P, P, c_size_t, P]),
    "dcf_knn_bev_sites": (c_int, [P, P, c_int, c_int, c_int, P, c_int, c_float, c_float, c_float, c_float, c_float, P]),
    "dcf_knn_bev_batch_shared": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, P, P, c_size_t, P, c_size_t, P]),
    "dcf_nchw_to_nhwc": (c_int, [c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "dcf_nhwc_to_nchw": (c_int, [c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "dcf_image_to_nhwc4": (c_int, [c_int, P, P, c_int, c_int, c_int, P]),
    "dcf_conv2d_fwd": (c_int, [c_int, P, P, P, P, P] + [c_int] * 12 + [P]),
    "dcf_conv2d_fwd_rowscale": (c_int, [c_int, P, P, P, P, P, P] + [c_int] * 12 + [P]),
    "dcf_conv2d_dgrad": (c_int, [c_int, P, P, P, P, P] + [c_int] * 11 + [P]),
    "dcf_conv2d_dgrad_halfres": (c_int, [c_int, P, P, P, P, P, P] + [c_int] * 11 + [P]),
    "dcf_conv3x3_chain_supported": (c_int, [c_int] * 6),
    "dcf_conv3x3_chain_workspace_bytes": (c_size_t, [c_int] * 6),
    "dcf_conv3x3_chain": (c_int, [c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "dcf_fp8_act_scale": (c_int, [c_float, P]),
    "dcf_cast_fp8": (c_int, [c_int, P, P, P, P, c_i64, P]),
    "dcf_weight_prep_fp8": (c_int, [P, P, c_int, c_int, P, P, P, P, P, c_float, P]),
    "dcf_conv2d_fwd_fp8": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P] + [c_int] * 12 + [P]),
    "dcf_conv2d_wgrad_splits": (c_int, [c_int] * 8),
    "dcf_conv2d_wgrad": (c_int, [c_int, P, P, P, P, c_int] + [c_int] * 11 + [P]),
    "dcf_conv2d_wgrad_groupable": (c_int, [c_int] * 10),
    "dcf_conv2d_wgrad_group": (c_int, [P, c_int, P]),
    "dcf_stem7x7_fwd": (c_int, [c_int, P, P, P, P] + [c_int] * 7 + [P]),
    "dcf_stem7x7_wgrad": (c_int, [c_int, P, P, P, P, c_int] + [c_int] * 6 + [P]),
    "dcf_weight_prep": (c_int, [c_int, P, c_int, P, P, P, P, c_float, P]),
    "dcf_wgrad_finalize": (c_int, [P, c_int, c_int, P, P, P, P, P, P, c_float, P]),
    "dcf_eval_score_filter": (c_int, [P, c_int, c_int, c_int, c_float, c_int, P, P, P]),
    "dcf_eval_nms_workspace_bytes": (c_size_t, [c_int]),
    "dcf_eval_nms": (c_int, [P, P, c_int, c_int, ctypes.c_double, P, P, P, P]),
    "dcf_eval_match": (c_int, [P, c_int, P, c_int, P, c_int, P, P]),
    "dcf_wgrad_finalize_rows": (c_int, [P, c_int, P, P, P, P, P, P, P, c_float, P]),
    "dcf_bn_workspace_bytes": (c_size_t, [c_int]),
    "dcf_bn_train_fwd": (c_int, [c_int, P, P, P, P, P, P, P, P, P, c_i64, c_int, c_float, c_float, c_int, P, P]),
    "dcf_bn_train_bwd": (c_int, [c_int, P, P, P, P, P, P, P, P, c_i64, c_int, P, P]),
    "dcf_relu_bwd_chansum": (c_int, [c_int, P, P, P, c_i64, c_int, c_int, P]),
    "dcf_resize_bilinear_fwd": (c_int, [c_int, P, P, P] + [c_int] * 7 + [P]),
    "dcf_resize_bilinear_bwd": (c_int, [c_int, P, P] + [c_int] * 7 + [P]),
    "dcf_maxpool3x3s2_fwd": (c_int, [c_int, P, P] + [c_int] * 6 + [P]),
    "dcf_maxpool3x3s2_bwd": (c_int, [c_int, P, P, P, P] + [c_int] * 6 + [P]),
    "dcf_maxpool3x3s2_fwd_idx": (c_int, [c_int, P, P, P] + [c_int] * 6 + [P]),
    "dcf_maxpool3x3s2_bwd_idx": (c_int, [c_int, P, P, P] + [c_int] * 6 + [P]),
    "dcf_head_fwd": (c_int, [c_int, P, c_int, P, P, c_int, c_int, c_int, P]),
    "dcf_head_bwd": (c_int, [c_int, P, c_int, P, P, P, P, c_int, c_int, c_int, P]),
    "dcf_point_sample_fwd": (c_int, [c_int, P, c_int, c_int, c_int, P, P, c_int, P, P]),
    "dcf_point_sample_bwd": (c_int, [c_int, P, c_int, c_int, c_int, P, P, c_int, P, P]),
    "dcf_point_sample_fwd_batch": (c_int, [c_int, P, c_int, c_int, c_int, P, c_i64, P, c_int, P, c_int, P]),
    "dcf_point_sample_bwd_batch": (c_int, [c_int, P, c_int, c_int, c_int, P, c_i64, P, c_int, P, c_int, P]),
    "dcf_fusion_gather_fwd_batch": (c_int, [c_int, P, c_i64, P, c_i64, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, c_int, P, P, c_int, P]),
    "dcf_fusion_gather_bwd_inv_batch": (c_int, [c_int, P, c_i64, P, c_i64, P, P, c_i64, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, c_int, P, P, P, P, P, c_int, P]),
    "dcf_fusion_gather_fwd": (c_int, [c_int, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, c_int, P, P, P]),
    "dcf_fusion_gather_bwd": (c_int, [c_int, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, c_int, P, P, P, P, P]),
    "dcf_fusion_invert_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dcf_fusion_invert": (c_int, [P, c_int, c_int, c_int, P, P, P, P, P]),
    "dcf_fusion_gather_bwd_pts": (c_int, [c_int, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, c_int, P, P, P, P, P]),
    "dcf_fusion_gather_bwd_workspace_bytes": (c_size_t, [c_int]),
    "dcf_fusion_gather_bwd_direct_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dcf_fusion_gather_bwd_direct_batch": (c_int, [c_int, P, c_i64, P, c_i64, P, P, c_i64, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, c_int, P, P, P, P, P, P, c_int, P]),
    "dcf_fusion_gather_bwd_inv": (c_int, [c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, P, P, c_int, P, P, P, P, P, P]),
    "dcf_rowscale_bias_fwd": (c_int, [c_int, P, P, P, c_i64, c_int, P]),
    "dcf_rowscale_bias_bwd": (c_int, [c_int, P, P, P, c_i64, c_int, P]),
    "dcf_relu_mask_rowscale_bwd": (c_int, [c_int, P, P, P, P, P, c_i64, c_int, P]),
    "dcf_cast": (c_int, [c_int, P, c_int, P, c_i64, P]),
    "dcf_loss_fwd_bwd": (c_int, [P, c_i64, P, c_i64, P, P, P, c_int, c_int, c_float, c_int, P, P, c_i64, P, c_i64, P]),
    "dcf_loss_sample_rand": (ctypes.c_uint32, [ctypes.c_uint64, c_int, c_int, c_int, c_int]),
    "dcf_loss_sample_fwd_bwd": (c_int, [P, c_i64, P, c_i64, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float,
                                        c_int, c_int, c_int, c_int, ctypes.c_uint64, c_float, c_int, P, P, c_i64, P, c_i64, P, P, P, P]),
    "dcf_adam_step": (c_int, [P, P, P, P, c_i64, c_float, c_float, c_float, c_float, c_int, c_float, P]),
}


class ConvParam(ctypes.Structure):
    """struct dcf_conv_param of include/dcf_hip.h."""
    _fields_ = [("w_off", c_i64), ("gamma_off", c_i64), ("beta_off", c_i64), ("mean_off", c_i64), ("var_off", c_i64),
                ("wfwd_off", c_i64), ("wdgrad_off", c_i64), ("shift_off", c_i64), ("slab_off", c_i64), ("gsum_off", c_i64),
                ("cout", ctypes.c_int32), ("cin", ctypes.c_int32), ("taps", ctypes.c_int32), ("cout_pad", ctypes.c_int32),
                ("nsplit", ctypes.c_int32), ("flags", ctypes.c_int32), ("pad0", ctypes.c_int32), ("pad1", ctypes.c_int32)]


F8_AMAX_STRIDE = 80          # DCF_F8_AMAX_STRIDE: 64 partial maxima + the previous step's maximum (+ padding) per conv


class F8Param(ctypes.Structure):
    """struct dcf_f8_param of include/dcf_hip.h."""
    _fields_ = [("w8_off", c_i64), ("wscale_off", c_i64)]


class WgradItem(ctypes.Structure):
    """struct dcf_wgrad_item of include/dcf_hip.h."""
    _fields_ = [("dtype", ctypes.c_int32), ("nsplit", ctypes.c_int32), ("x", c_void_p), ("gy", c_void_p), ("slabs", c_void_p),
                ("gsum", c_void_p), ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Cin", ctypes.c_int32),
                ("Cout", ctypes.c_int32), ("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("pad_", ctypes.c_int32)]


CHAIN_MAX_LAYERS = 24        # DCF_CHAIN_MAX_LAYERS


class ChainLayer(ctypes.Structure):
    """struct dcf_chain_layer of include/dcf_hip.h."""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("shift", c_void_p), ("res", c_void_p), ("mask", c_void_p), ("y", c_void_p),
                ("relu", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class KnnMap(ctypes.Structure):
    """struct dcf_knn_map of include/dcf_hip.h."""
    _fields_ = [("idx", c_void_p), ("h", ctypes.c_int32), ("w", ctypes.c_int32)]


class KnnSite(ctypes.Structure):
    """struct dcf_knn_site of include/dcf_hip.h."""
    _fields_ = [("h", ctypes.c_int32), ("w", ctypes.c_int32), ("stride", ctypes.c_int32), ("fine", ctypes.c_int32), ("idx_out", c_void_p),
                ("ws", c_void_p), ("ws_stride_bytes", ctypes.c_size_t)]


class DcfError(RuntimeError):
    pass


_LIB = None


def build(force=False):
    """Compile libdcf_hip.so for gfx950 with hipcc (in-tree, see csrc/Makefile)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    return LIB_PATH


def lib():
    """Load the shared library; raises (no fallback) when it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DcfError("libdcf_hip.so is not built (%s). Run __graft_entry__.build() / make -C %s; "
                           "there is no CPU fallback for the HIP hot path." % (LIB_PATH, CSRC))
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


_Tensor = torch.Tensor
_Param = torch.nn.Parameter


def _ptr(t):
    # hot path (~20 arguments per launch, ~230 launches per step): exact-type tests first, isinstance only for the rest
    tt = type(t)
    if tt is int or tt is float or t is None:
        return t
    if tt is _Tensor or tt is _Param:
        return t.data_ptr()
    if isinstance(t, torch.Tensor):
        return t.data_ptr()
    if isinstance(t, np.ndarray):
        return t.ctypes.data
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else None


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (every launch goes onto it)."""
    if _raw_stream is not None:          # one C call instead of torch.cuda.current_stream()'s Python layers (~7 us, x600 per step)
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


_FN = {}                # name -> (bound foreign function, raises on a non-zero status)
_NO_RAISE = ("dcf_version", "dcf_conv2d_wgrad_splits", "dcf_prof_read", "dcf_conv2d_wgrad_groupable", "dcf_conv3x3_chain_supported")


def call(name, *args):
    """Invoke dcf_<name>; tensors/ndarrays become raw pointers; non-zero status raises."""
    ent = _FN.get(name)
    if ent is None:
        fn = getattr(lib(), name)
        ent = _FN[name] = (fn, fn.restype is c_int and name not in _NO_RAISE)
    rc = ent[0](*[_ptr(a) for a in args])
    if rc != 0 and ent[1]:
        raise DcfError("%s failed (%d): %s" % (name, rc, lib().dcf_last_error().decode()))
    return rc


def fn(name):
    """The bound foreign function itself, for call sites hot enough to marshal their own arguments (ints / None only); a
    non-zero status goes to fail()."""
    return getattr(lib(), name)


def fail(name, rc):
    raise DcfError("%s failed (%d): %s" % (name, rc, lib().dcf_last_error().decode()))


def set_option(name, value):
    """Tuning option of the library (dcf_set_option): which kernel / tile shape a launch takes, never its result.
    value None = unset (back to the built-in choice)."""
    rc = lib().dcf_set_option(name.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise DcfError("dcf_set_option(%r) failed: %s" % (name, lib().dcf_last_error().decode()))


def dtype_code(dt):
    if dt in (torch.float32, "f32", "fp32", F32):
        return F32
    if dt in (torch.bfloat16, "bf16", BF16):
        return BF16
    if dt in (torch.float16, "f16", "fp16", F16):
        return F16
    raise DcfError("unsupported compute dtype %r (f32, bf16 or f16)" % (dt,))


def torch_dtype(code):
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[code]


def host_f32(values):
    return np.ascontiguousarray(np.asarray(values, dtype=np.float32))


def prof_read(cap=256):
    names = ctypes.create_string_buffer(cap * 64)
    tot = (ctypes.c_double * cap)()
    cnt = (ctypes.c_int64 * cap)()
    wk = (ctypes.c_double * cap)()
    by = (ctypes.c_double * cap)()
    k = lib().dcf_prof_read2(ctypes.cast(names, c_void_p), ctypes.cast(tot, c_void_p), ctypes.cast(cnt, c_void_p),
                             ctypes.cast(wk, c_void_p), ctypes.cast(by, c_void_p), cap)
    out = {}
    for i in range(k):
        nm = names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode()
        out[nm] = (tot[i], cnt[i], wk[i], by[i])       # (total ms, launches, algorithmic flops, algorithmic bytes)
    return out
