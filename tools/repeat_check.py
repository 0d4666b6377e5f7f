#!/usr/bin/env python3
"""Which convolution launch is not bitwise repeatable under load?  Every cfg2 layer shape, forward and input gradient (and the
weight gradient), fp32 / bf16, N times while a second stream moves 512 MB back and forth; prints the shapes whose output bits
changed between runs.  Usage (GPU box): python tools/repeat_check.py [--dtype f32] [--runs 40] [--batch 2]"""
import argparse, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE
ops = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.ops")
H = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd._hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--runs", type=int, default=40)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--no-hog", action="store_true")
    ap.add_argument("--sibling", action="store_true", help="a second process (bench.py) trains on the same GPU meanwhile")
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    sib = None
    if args.sibling:
        import subprocess
        sib = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1500", "--warmup", "2", "--no-cpu-baseline", "--no-roofline",
                                "--no-other-leg", "--no-batch-sweep", "--input", "resident"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    dt = {"f32": 0, "bf16": 1, "f16": 2}[args.dtype]
    td = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}[dt]
    B = args.batch
    hs = torch.cuda.Stream()
    ha = torch.empty(128 << 20, dtype=torch.float32, device="cuda"); hb = torch.empty_like(ha)
    extra = [("stem", 375, 1242, 3, 64, 7, 2, 1)]
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if args.names and name not in args.names:
            continue
        pad = k // 2
        Ho, Wo = (Hh + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).to(td)
        w = ((torch.rand((Co, k, k, Ci), device="cuda") - 0.5) * 0.1).to(td)
        wt = w.permute(3, 1, 2, 0).contiguous()
        gy = (torch.rand((B, Ho, Wo, Co), device="cuda") - 0.5).to(td)
        res = (torch.rand((B, Ho, Wo, Co), device="cuda") - 0.5).to(td)
        shift = torch.rand((Co,), device="cuda") - 0.5
        ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Ci, Co, k, k, s)
        slabs = torch.empty((ns, Co, k, k, Ci), device="cuda")
        ref = None
        bad = {"fwd": 0, "dgrad": 0, "wgrad": 0}
        for run in range(args.runs + 1):
            if run > 0 and not args.no_hog:
                with torch.cuda.stream(hs):
                    for _ in range(2):
                        hb.copy_(ha, non_blocking=True); ha.copy_(hb, non_blocking=True)
            yf = ops.conv2d_fwd(dt, x, w, shift, res, k, k, s, pad, True, Co)
            yd = ops.conv2d_dgrad(dt, gy, wt, None, (B, Hh, W, Ci), k, k, s, pad)
            ops.conv2d_wgrad(dt, x, gy, slabs, ns, k, k, s, pad)
            torch.cuda.synchronize()
            cur = (yf.clone(), yd.clone(), slabs.clone())
            if ref is None:
                ref = cur
                continue
            for key, a, b in zip(("fwd", "dgrad", "wgrad"), cur, ref):
                if not torch.equal(a, b):
                    bad[key] += 1
        flag = "  <-- NOT REPEATABLE" if any(bad.values()) else ""
        if sib is not None and sib.poll() is not None:
            print("(the sibling process has ended)")
        print("%-6s %4dx%-4d %3d->%3d k%d s%d  runs differing: fwd %d dgrad %d wgrad %d of %d%s" % (name, Hh, W, Ci, Co, k, s, bad["fwd"], bad["dgrad"], bad["wgrad"], args.runs, flag), flush=True)
    if sib is not None:
        sib.terminate(); sib.wait(timeout=60)


if __name__ == "__main__":
    main()
