#!/usr/bin/env python3
"""Loader / consumer row-sharing kernel (conv_lc.hip, the default) against conv_rs.hip (option CONV_LC=0) on the cfg2 layer shapes:
bit-equality of the outputs (forward with shift + residual + ReLU, input gradient with residual + mask) and time per launch.
Usage (GPU box): python tools/lc_bench.py [--batch 2] [names...]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE, timeit
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")
H = importlib.import_module(PKG + "._hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    dt = 1 if args.dtype == "bf16" else 2
    td = torch.bfloat16 if dt == 1 else torch.float16
    B = args.batch
    want = set(args.names)
    tot = {"rs_fwd": 0.0, "lc_fwd": 0.0, "rs_dgrad": 0.0, "lc_dgrad": 0.0}
    print("%-6s %4s %4s %4s %4s | fwd rs us  lc us  TF/s(lc) eq | dgrad rs us  lc us  TF/s(lc) eq" % ("name", "H", "W", "Ci", "Co"))
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if k != 3 or s != 1 or Ci % 64 or Co % 64 or (want and name not in want):
            continue
        torch.manual_seed(1)
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).to(td)
        w = ((torch.rand((Co, 3, 3, Ci), device="cuda") - 0.5) * 0.1).to(td)
        wt = w.permute(3, 1, 2, 0).contiguous()
        gy = (torch.rand((B, Hh, W, Co), device="cuda") - 0.5).to(td)
        shift = torch.rand((Co,), device="cuda") - 0.5
        res = (torch.rand((B, Hh, W, Co), device="cuda") - 0.5).to(td)
        resg = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).to(td)
        mask = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.3).to(td)
        H.set_option("CONV_LC", 0)
        y0 = ops.conv2d_fwd(dt, x, w, shift, res, 3, 3, 1, 1, True, Co)
        g0 = ops.conv2d_dgrad(dt, gy, wt, resg, (B, Hh, W, Ci), 3, 3, 1, 1, mask=mask)
        H.set_option("CONV_LC", None)
        y1 = ops.conv2d_fwd(dt, x, w, shift, res, 3, 3, 1, 1, True, Co)
        g1 = ops.conv2d_dgrad(dt, gy, wt, resg, (B, Hh, W, Ci), 3, 3, 1, 1, mask=mask)
        torch.cuda.synchronize()
        # (the 2-D kernel sums chunk-major, conv_rs kernel-row-major: the same products in another fp32 order -- an output may land
        # on the neighbouring 16-bit value)
        eqf, eqd = bool(torch.allclose(y0.float(), y1.float(), atol=1.6e-2, rtol=0)), bool(torch.allclose(g0.float(), g1.float(), atol=1.6e-2, rtol=0))
        if not eqf:
            d = (y0.float() - y1.float()).abs()
            print("   fwd mismatch: max %.4g, %d of %d elements, first at %s" % (d.max().item(), int((d > 0).sum()), d.numel(), tuple((d > 0).nonzero()[0].tolist())))
        if not eqd:
            d = (g0.float() - g1.float()).abs()
            print("   dgrad mismatch: max %.4g, %d of %d elements, first at %s" % (d.max().item(), int((d > 0).sum()), d.numel(), tuple((d > 0).nonzero()[0].tolist())))
        fl = 2.0 * B * Hh * W * Co * Ci * 9
        H.set_option("CONV_LC", 0)
        t0 = timeit(lambda: ops.conv2d_fwd(dt, x, w, None, None, 3, 3, 1, 1, False, Co))
        t2 = timeit(lambda: ops.conv2d_dgrad(dt, gy, wt, resg, (B, Hh, W, Ci), 3, 3, 1, 1, mask=mask))
        H.set_option("CONV_LC", None)
        t1 = timeit(lambda: ops.conv2d_fwd(dt, x, w, None, None, 3, 3, 1, 1, False, Co))
        t3 = timeit(lambda: ops.conv2d_dgrad(dt, gy, wt, resg, (B, Hh, W, Ci), 3, 3, 1, 1, mask=mask))
        tot["rs_fwd"] += t0 * cnt; tot["lc_fwd"] += t1 * cnt; tot["rs_dgrad"] += t2 * cnt; tot["lc_dgrad"] += t3 * cnt
        print("%-6s %4d %4d %4d %4d | %8.1f %6.1f %8.0f %s | %10.1f %6.1f %8.0f %s  (x%d)" % (
            name, Hh, W, Ci, Co, t0 * 1e6, t1 * 1e6, fl / t1 / 1e12, ("%.0e" % (y0.float() - y1.float()).abs().max().item()) if eqf else "NE", t2 * 1e6, t3 * 1e6, fl / t3 / 1e12, ("%.0e" % (g0.float() - g1.float()).abs().max().item()) if eqd else "NE", cnt), flush=True)
    print("weighted totals (ms):", {k: round(v * 1e3, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
