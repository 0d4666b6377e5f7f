#!/usr/bin/env python3
"""Where the row-sharing convolution kernel's time goes: phases switched off one at a time (option RS_DBG; results are wrong
in those runs, only the clock counts).  Needs the ablation build of the library -- the shipped one has no such switch:
    make -C <pkg>/csrc clean && make -C <pkg>/csrc ABLATE=1 && python tools/rs_ablate.py [names...]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE, timeit
ops = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.ops")
H = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd._hip")
MODES = [(0, "as shipped"), (1, "no MFMAs"), (2, "no pixel fetch"), (16, "no weight fetch"), (18, "no fetch at all"),
         (4, "no epilogue"), (19, "no MFMA, no fetch"), (23, "DMA issue + barriers only")]


def main():
    want = set(sys.argv[1:]) or {"l2", "l3", "l4", "l5"}
    B = 2
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if name not in want:
            continue
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).bfloat16()
        w = ((torch.rand((Co, 3, 3, Ci), device="cuda") - 0.5) * 0.1).bfloat16()
        out = []
        for dbg, label in MODES:
            H.set_option("RS_DBG", dbg)
            t = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, Co), iters=20)
            out.append("%s %.1f" % (label, t * 1e6))
        H.set_option("RS_DBG", None)
        print("%-6s %s" % (name, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
