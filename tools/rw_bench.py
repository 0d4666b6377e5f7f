#!/usr/bin/env python3
"""Register-weight row-sharing kernel (conv_rw.hip) against the LDS-staged one (conv_rs.hip) on the cfg2 layer shapes:
bit-equality of the outputs (forward with shift + residual + ReLU, input gradient with residual + mask) and time per launch.
Usage (GPU box): python tools/rw_bench.py [--batch 2] [names...]"""
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
    tot = {"rs_fwd": 0.0, "rw_fwd": 0.0, "rs_dgrad": 0.0, "rw_dgrad": 0.0}
    print("%-6s %4s %4s %4s %4s | fwd rs us  rw us  TF/s(rw) eq | dgrad rs us  rw us  TF/s(rw) eq" % ("name", "H", "W", "Ci", "Co"))
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
        wf = rw_api.conv3x3_weight_frag(dt, w)
        wtf = rw_api.conv3x3_weight_frag(dt, wt)
        y0 = ops.conv2d_fwd(dt, x, w, shift, res, 3, 3, 1, 1, True, Co)
        y1 = rw_api.conv3x3_fwd_wf(dt, x, wf, shift, res, True, Co)
        g0 = ops.conv2d_dgrad(dt, gy, wt, resg, (B, Hh, W, Ci), 3, 3, 1, 1, mask=mask)
        g1 = rw_api.conv3x3_dgrad_wf(dt, gy, wtf, resg, (B, Hh, W, Ci), mask=mask)
        torch.cuda.synchronize()
        eqf, eqd = bool(torch.equal(y0, y1)), bool(torch.equal(g0, g1))
        if not eqf:
            d = (y0.float() - y1.float()).abs()
            print("   fwd mismatch: max %.4g, %d of %d elements, first at %s" % (d.max().item(), int((d > 0).sum()), d.numel(), tuple((d > 0).nonzero()[0].tolist())))
        if not eqd:
            d = (g0.float() - g1.float()).abs()
            print("   dgrad mismatch: max %.4g, %d of %d elements, first at %s" % (d.max().item(), int((d > 0).sum()), d.numel(), tuple((d > 0).nonzero()[0].tolist())))
        fl = 2.0 * B * Hh * W * Co * Ci * 9
        t0 = timeit(lambda: ops.conv2d_fwd(dt, x, w, None, None, 3, 3, 1, 1, False, Co))
        t1 = timeit(lambda: rw_api.conv3x3_fwd_wf(dt, x, wf, None, None, False, Co))
        t2 = timeit(lambda: ops.conv2d_dgrad(dt, gy, wt, resg, (B, Hh, W, Ci), 3, 3, 1, 1, mask=mask))
        t3 = timeit(lambda: rw_api.conv3x3_dgrad_wf(dt, gy, wtf, resg, (B, Hh, W, Ci), mask=mask))
        tot["rs_fwd"] += t0 * cnt; tot["rw_fwd"] += t1 * cnt; tot["rs_dgrad"] += t2 * cnt; tot["rw_dgrad"] += t3 * cnt
        print("%-6s %4d %4d %4d %4d | %8.1f %6.1f %8.0f %s | %10.1f %6.1f %8.0f %s  (x%d)" % (
            name, Hh, W, Ci, Co, t0 * 1e6, t1 * 1e6, fl / t1 / 1e12, "==" if eqf else "NE", t2 * 1e6, t3 * 1e6, fl / t3 / 1e12, "==" if eqd else "NE", cnt), flush=True)
    print("weighted totals (ms):", {k: round(v * 1e3, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
