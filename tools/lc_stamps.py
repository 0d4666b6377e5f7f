#!/usr/bin/env python3
"""Barrier-by-barrier timeline of one k_conv3x3_lc launch (needs a -DLC_STAMP build of conv_lc.hip:
    KFILE=conv_lc bash tools/rw_variants.sh stamp="-DLC_STAMP"; DCF_HIP_LIB=<pkg>/libdcf_hip_vstamp.so python tools/lc_stamps.py l3 [--batch 2]).
Per barrier g of workgroup 0: when each wave arrived (cycles since the workgroup's first stamp) and when the barrier opened:
the wave that arrives last is the one everybody waited for.  Waves 0-3 consumers, 4-5 weight loaders, 6-7 pixel loaders."""
import argparse, ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conv_bench import LIDAR, IMAGE
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")
H = importlib.import_module(PKG + "._hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--wg", type=int, default=0)
    ap.add_argument("--summary", action="store_true", help="one line: median busy cycles per tap of each wave")
    ap.add_argument("name")
    args = ap.parse_args()
    L = ctypes.CDLL(H.LIB_PATH)
    B = args.batch
    for name, Hh, W, Ci, Co, k, s, cnt in LIDAR + IMAGE:
        if name != args.name:
            continue
        x = (torch.rand((B, Hh, W, Ci), device="cuda") - 0.5).bfloat16()
        w = ((torch.rand((Co, 3, 3, Ci), device="cuda") - 0.5) * 0.1).bfloat16()
        for _ in range(3):
            ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, Co)
        torch.cuda.synchronize()
        L.dcf_lc_stamps_clear()
        ops.conv2d_fwd(1, x, w, None, None, 3, 3, 1, 1, False, Co)
        torch.cuda.synchronize()
        dims = (ctypes.c_int * 4)()
        buf = np.zeros(4 * 8 * 160 * 2, dtype=np.int64)
        L.dcf_lc_stamps_read(buf.ctypes.data_as(ctypes.c_void_p), dims)
        st = buf.reshape(dims[0], dims[1], dims[2], dims[3])[args.wg]
        t0 = st[st > 0].min()
        marks = st[:, dims[2] - 8:, 0]
        if (marks > 0).any():
            for wv in range(8):
                if (marks[wv] > 0).any():
                    print("wave %d prologue marks (cycles since first stamp): %s" % (wv, " ".join("%d" % (m - t0) for m in marks[wv] if m > 0)))
        st = st.copy(); st[:, dims[2] - 8:, :] = 0
        ng = int((st[0, :, 0] > 0).sum())
        if args.summary:
            busy = np.zeros((8, ng - 1))
            for g in range(1, ng):
                busy[:, g - 1] = st[:, g, 0] - st[:, g - 1, 1].min()
            print("TH/TW %s/%s: barrier 0 opened at %d; median busy cycles per tap, waves 0..7: %s; median period %d" % (
                os.environ.get("DCF_LC_TH"), os.environ.get("DCF_LC_TW"), int(st[:, 0, 1].min() - t0), " ".join("%5d" % v for v in np.median(busy, axis=1)),
                int(np.median(np.diff(st[:, :ng, 1].min(axis=0))))))
            return
        print("barrier | arrival of waves 0..7 (cycles)                                  | opened | last to arrive | since previous")
        prev = 0
        for g in range(ng):
            arr = st[:, g, 0] - t0
            opened = int((st[:, g, 1] - t0).min())
            last = int(np.argmax(arr))
            print("%7d | %s | %6d | wave %d (%s) | %6d" % (g, " ".join("%7d" % a for a in arr), opened, last,
                  "consumer" if last < 4 else ("W loader" if last < 6 else "X loader"), opened - prev))
            prev = opened


if __name__ == "__main__":
    main()
