#!/usr/bin/env python3
"""us per launch of the row-sharing convolution on given shapes (BxHxWxC ...), forward (plain epilogue) -- for variant libraries
(DCF_HIP_LIB=... python tools/rs_time.py 2x88x100x192 2x176x200x128); results of ablation variants are wrong by design."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import timeit
ops = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.ops")
out = []
for shp in sys.argv[1:]:
    B, Hh, W, C = [int(v) for v in shp.split("x")]
    x = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
    w = ((torch.rand((C, 3, 3, C), device="cuda") - 0.5) * 0.05).bfloat16()
    t = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, True, C), iters=50)
    out.append("%s %.1f" % (shp, t * 1e6))
print(os.path.basename(os.environ.get("DCF_HIP_LIB", "product")), " | ".join(out))
