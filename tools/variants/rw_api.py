"""ctypes bindings of the experimental register-weight convolution (tools/variants/conv_rw.hip): only a variant library
built by `bash tools/rw_variants.sh <name>=<flags>` exports these symbols (DCF_HIP_LIB=<pkg>/libdcf_hip_v<name>.so); the
shipped libdcf_hip.so does not."""
import importlib
import os
import sys
from ctypes import c_int, c_void_p

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
H = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd._hip")
P = c_void_p
SIGS = {
    "dcf_conv3x3_wf_supported": (c_int, [c_int] * 6),
    "dcf_conv3x3_weight_frag": (c_int, [c_int, P, P, c_int, c_int, P]),
    "dcf_conv3x3_fwd_wf": (c_int, [c_int, P, P, P, P, P] + [c_int] * 6 + [P]),
    "dcf_conv3x3_dgrad_wf": (c_int, [c_int, P, P, P, P, P] + [c_int] * 5 + [P]),
}


def _fn(name):
    lib = H.lib()
    try:
        f = getattr(lib, name)
    except AttributeError:
        raise H.DcfError("%s is not in %s: build a variant library with tools/rw_variants.sh and point DCF_HIP_LIB at it" % (name, lib._name))
    f.restype, f.argtypes = SIGS[name]
    return f


def _p(t):
    return None if t is None else (t if isinstance(t, int) else t.data_ptr())


def _call(name, *a):
    rc = _fn(name)(*a)
    if rc:
        H.fail(name, rc)


def conv3x3_weight_frag(dtype, w):
    Cout, kh, kw, Cin = w.shape
    wf = torch.empty_like(w)
    _call("dcf_conv3x3_weight_frag", dtype, _p(w), _p(wf), Cout, Cin, H.stream_ptr())
    return wf


def conv3x3_fwd_wf(dtype, x, wf, shift, res, relu, cout):
    B, Hh, W, Cin = x.shape
    y = torch.empty((B, Hh, W, cout), dtype=x.dtype, device=x.device)
    _call("dcf_conv3x3_fwd_wf", dtype, _p(x), _p(wf), _p(shift), _p(res), _p(y), B, Hh, W, Cin, cout, int(relu), H.stream_ptr())
    return y


def conv3x3_dgrad_wf(dtype, gy, wtf, res, in_shape, mask=None):
    B, Hh, W, Cin = in_shape
    Cout = gy.shape[3]
    gx = torch.empty((B, Hh, W, Cin), dtype=gy.dtype, device=gy.device)
    _call("dcf_conv3x3_dgrad_wf", dtype, _p(gy), _p(wtf), _p(res), _p(mask), _p(gx), B, Hh, W, Cin, Cout, H.stream_ptr())
    return gx
