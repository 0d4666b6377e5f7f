#!/usr/bin/env python3
"""Where the spatial-tile kernel's time goes (conv_sp.hip): phases switched off (option SP_DBG; wrong results, only the clock
counts), batch sizes, rows per tile.  Needs the ablation build:  make -C <pkg>/csrc clean; make -C <pkg>/csrc ABLATE=1
Usage (GPU box): python tools/sp_ablate.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import timeit
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")
H = importlib.import_module(PKG + "._hip")
MODES = [(0, "as shipped"), (1, "no MFMA/LDS reads"), (2, "DMA reads nothing"), (4, "no stores"), (3, "stores only"), (5, "DMA only"), (6, "MFMA only"), (7, "nothing")]


def main():
    H.set_option("CONV_SP_MIN_PIX", 0)
    for name, Hh, W, C, B in (("l1", 704, 800, 32, 2), ("l2", 352, 400, 64, 2), ("l2", 352, 400, 64, 8)):
        x = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
        w = ((torch.rand((C, 3, 3, C), device="cuda") - 0.5) * 0.1).bfloat16()
        for th, wg in ((4, 2), (4, 1)) if C == 64 else ((8, 2), (8, 1)):
            H.set_option("CONV_SP_WGPC", wg)
            out = []
            for dbg, label in MODES:
                H.set_option("SP_DBG", dbg)
                t = timeit(lambda: ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, C), iters=20)
                out.append("%s %.1f" % (label, t * 1e6))
            H.set_option("SP_DBG", None)
            print("%s C=%d B=%d TH=%d wg/CU=%d: %s" % (name, C, B, th, wg, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
