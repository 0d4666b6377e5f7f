#!/usr/bin/env python3
"""Where the register-weight row-sharing kernel's time goes: phases switched off one at a time (option RW_DBG; results are
wrong in those runs, only the clock counts).  Needs the ablation build:
    make -C <pkg>/csrc ablate && DCF_HIP_LIB=<pkg>/libdcf_hip_ablate.so python tools/rw_ablate.py [--batch 2] [names...]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE, timeit
sys.path.insert(0, os.path.join(ROOT, "tools", "variants"))
import rw_api          # the register-weight kernel lives in a variant library (tools/rw_variants.sh), not in libdcf_hip.so
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")
H = importlib.import_module(PKG + "._hip")
MODES = [(0, "as shipped"), (1, "no MFMAs"), (2, "no pixel fetch"), (16, "no weight fetch"), (18, "no fetch at all"),
         (4, "no epilogue"), (5, "no MFMA, no epilogue"), (22, "no fetch, no epilogue"), (23, "issue + barriers only"),
         (23 + 32, "... pixel DMA not issued"), (23 + 64, "... weight loads not issued"), (23 + 96, "... neither issued"),
         (23 + 128, "... no LDS reads"), (23 + 96 + 128, "... barriers only"), (4 + 96, "MFMA + LDS reads only"), (4 + 96 + 128, "MFMA only")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    want = set(args.names) or {"l3", "l4", "l5"}
    B = args.batch
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if name not in want or k != 3 or s != 1:
            continue
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).bfloat16()
        w = ((torch.rand((Co, 3, 3, Ci), device="cuda") - 0.5) * 0.1).bfloat16()
        wf = rw_api.conv3x3_weight_frag(1, w)
        out = []
        for dbg, label in MODES:
            H.set_option("RW_DBG", dbg)
            t = timeit(lambda: rw_api.conv3x3_fwd_wf(1, x, wf, None, None, False, Co), iters=30)
            out.append("%s %.1f" % (label, t * 1e6))
        H.set_option("RW_DBG", None)
        print("%-6s %s" % (name, " | ".join(out)), flush=True)


if __name__ == "__main__":
    main()
