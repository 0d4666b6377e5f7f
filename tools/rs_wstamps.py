#!/usr/bin/env python3
"""Per-wave barrier stamps of one single-layer launch of the row-sharing convolution (needs a -DRS_WSTAMP build:
    KFILE=conv_rs bash tools/rw_variants.sh wstamp="-DRS_WSTAMP"
    DCF_HIP_LIB=<pkg>/libdcf_hip_vwstamp.so python tools/rs_wstamps.py 2x88x100x192 [--dgrad] [--opt RS_PF=0]).
Every wave of every workgroup stamps s_memtime when it ARRIVES at and when it is RELEASED from each tap barrier of its first
tile, plus six phase marks.  Printed: the phases (median over workgroups, us at the measured clock), and per tap barrier which
wave arrives last, how long the others have been idle there, and how long a tap takes from release to the next release."""
import argparse, ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")
H = importlib.import_module(PKG + "._hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", help="BxHxWxC")
    ap.add_argument("--dgrad", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="NAME=VALUE library options")
    ap.add_argument("--mhz", type=float, default=100.0, help="ticks per printed unit: s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md), "
                    "so the default prints in units of 100 cycles (~0.05 us at the ~2.05 GHz the chip holds under this load)")
    args = ap.parse_args()
    for o in args.opt:
        k, v = o.split("=")
        H.set_option(k, v)
    B, Hh, W, C = [int(v) for v in args.shape.split("x")]
    L = ctypes.CDLL(H.LIB_PATH)
    x = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
    w = ((torch.rand((C, 3, 3, C), device="cuda") - 0.5) * 0.05).bfloat16()
    res = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
    mask = (torch.rand((B, Hh, W, C), device="cuda") - 0.3).bfloat16()

    def run():
        if args.dgrad:
            return ops.conv2d_dgrad(1, x, w, res, (B, Hh, W, C), 3, 3, 1, 1, mask=mask)
        return ops.conv2d_fwd(1, x, w, None, res, 3, 3, 1, 1, True, C)
    L.dcf_rs_wstamps_clear()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(30):
        run()
    ev0.record()
    for _ in range(20):
        run()
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / 20
    dims = (ctypes.c_int * 4)()
    buf = np.zeros(256 * 8 * 3 * 64, dtype=np.uint32)
    assert L.dcf_rs_wstamps_read(buf.ctypes.data_as(ctypes.c_void_p), dims) == 0
    st = buf.reshape(256, 8, 3, 64).astype(np.int64)
    live = [g for g in range(256) if st[g, 0, 2, 0] != 0]
    st = st[live]
    arr, rel, misc = st[:, :, 0, :], st[:, :, 1, :], st[:, :, 2, :]
    nb = int((arr[0, 0] != 0).sum())
    t0 = misc[:, :, 0].min(axis=1)                       # workgroup start = its first wave's entry
    tick = 1.0 / args.mhz                                # us per tick

    def d(a, b):                                         # wrapped 32-bit difference
        return ((a - b) & 0xFFFFFFFF).astype(np.float64) * tick
    print("%s %s: %.1f us per launch (back to back), %d workgroups stamped, %d barriers per tile; everything below in units of 100 shader cycles" % (args.shape, "dgrad" if args.dgrad else "fwd", us, len(live), nb))
    span = d(misc[:, :, 5].max(axis=1), t0)
    print("workgroup lifetime (entry of first wave -> stores drained, slowest wave): median %.2f  p90 %.2f  max %.2f us" % (np.median(span), np.percentile(span, 90), span.max()))
    names = ["entry -> first groups issued", "-> first barrier released (first data landed)", "main loop (first release -> last MFMA issued)",
             "DMA tail drained", "epilogue: loads + stores issued", "stores drained"]
    marks = [d(misc[:, :, 1], misc[:, :, 0]), d(rel[:, :, 0], misc[:, :, 1]), d(misc[:, :, 2], rel[:, :, 0]), d(misc[:, :, 3], misc[:, :, 2]),
             d(misc[:, :, 4], misc[:, :, 3]), d(misc[:, :, 5], misc[:, :, 4])]
    for n, m in zip(names, marks):
        print("  %-52s median %6.2f us   p90 %6.2f   (per wave, all workgroups)" % (n, np.median(m), np.percentile(m, 90)))
    # kernel-level, per XCD (every XCD has its own counter): first entry -> last drain over the XCD's workgroups
    xcd = np.array([g & 7 for g in live])
    spans, skews = [], []
    for xi in range(8):
        m = xcd == xi
        if not m.any():
            continue
        g0 = t0[m].min()
        skews.append(d(t0[m].max(), g0))
        spans.append(d(misc[m][:, :, 5].max(), g0))
    print("  per XCD: workgroup entries spread over %.2f (max %.2f); first entry -> last drain %.2f (max %.2f)" % (
        np.mean(skews), np.max(skews), np.mean(spans), np.max(spans)))
    # barriers
    print("tap barriers (median over workgroups): skew = last arrival - first arrival; idle = mean wait of a wave at the barrier; tap = release -> next release")
    last_hist = np.zeros(8, dtype=int)
    rows = []
    for t in range(nb):
        a, r = arr[:, :, t], rel[:, :, t]
        first, lastw = a.min(axis=1), a.argmax(axis=1)
        for wv in lastw:
            last_hist[wv] += 1
        skew = d(a.max(axis=1), first)
        idle = d(r, a).mean(axis=1)
        tap = d(rel[:, :, t + 1].max(axis=1), r.max(axis=1)) if t + 1 < nb else np.zeros(len(live))
        rows.append((t, np.median(skew), np.median(idle), np.median(tap)))
    for t, sk, idl, tp in rows:
        print("  tap %2d  skew %5.2f us   idle %5.2f us   tap %5.2f us" % (t, sk, idl, tp))
    print("  sum over taps: idle at barriers %.2f us of %.2f us main loop" % (sum(r[2] for r in rows), sum(r[3] for r in rows)))
    print("  last to arrive, by wave id (all taps, all workgroups): %s" % " ".join("w%d:%d" % (i, c) for i, c in enumerate(last_hist)))


if __name__ == "__main__":
    main()
