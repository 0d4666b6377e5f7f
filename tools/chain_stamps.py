#!/usr/bin/env python3
"""Where a tile of a CHAIN launch spends its time (needs a -DRS_STAMP build of conv_chain.hip:
    KFILE=conv_chain bash tools/rw_variants.sh stamp="-DRS_STAMP"
    DCF_HIP_LIB=<pkg>/libdcf_hip_vstamp.so python tools/chain_stamps.py 2x44x50x256 11 [--dgrad]).
Wave 0 of every workgroup stamps s_memtime at nine points of every work item (conv_rs_kernel.h, RS_STAMP); this prints, per
phase, the median / p90 over workgroups and layers, and the per-layer critical path (latest publish of layer l -> latest
publish of layer l + 1)."""
import argparse, ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"
ops = importlib.import_module(PKG + ".ops")
H = importlib.import_module(PKG + "._hip")
PH = ["wait for the previous layer's tiles", "issue first DMA groups", "first stage lands", "main loop (MFMAs)", "DMA tail drain",
      "epilogue (residual / mask loads, stores issued)", "store drain", "barrier + publish"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", help="BxHxWxC")
    ap.add_argument("layers", type=int)
    ap.add_argument("--dgrad", action="store_true")
    ap.add_argument("--plain", action="store_true", help="no residual / mask / shift epilogue")
    args = ap.parse_args()
    B, Hh, W, C = [int(v) for v in args.shape.split("x")]
    n = args.layers
    L = ctypes.CDLL(H.LIB_PATH)
    x = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
    ext = (torch.rand((B, Hh, W, C), device="cuda") - 0.5).bfloat16()
    wts = [((torch.rand((C, 3, 3, C), device="cuda") - 0.5) * 0.05).bfloat16() for _ in range(n)]
    masks = [(torch.rand((B, Hh, W, C), device="cuda") - 0.3).bfloat16() for _ in range(n)]
    ws = ops.conv3x3_chain_workspace(1, B, Hh, W, C, n, "cuda")
    if args.plain:
        layers = [(wts[l], None, None, None, False) for l in range(n)]
    elif args.dgrad:
        layers = [(wts[l], None, None if l % 2 == 0 else (ext if l == 1 else l - 2), masks[l], False) for l in range(n)]
    else:
        layers = [(wts[l], None, (ext if l == 0 else l - 2) if l % 2 == 0 else None, None, True) for l in range(n)]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.dcf_rs_stamps_clear()
    # (every launch overwrites the stamps: the LAST of a back-to-back series is read, with the clocks up and the caches warm --
    # a lone launch after the host-side clear runs at an idle chip's clock)
    for _ in range(30):
        ops.conv3x3_chain(1, x, layers, int(args.dgrad), ws)
    ev0.record()
    for _ in range(10):
        ops.conv3x3_chain(1, x, layers, int(args.dgrad), ws)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / 10
    assert ops.conv3x3_chain_status(ws) == 0
    dims = (ctypes.c_int * 3)()
    buf = np.zeros(256 * 32 * 10, dtype=np.int64)
    L.dcf_rs_stamps_read(buf.ctypes.data_as(ctypes.c_void_p), dims)
    st = buf.reshape(dims[0], dims[1], dims[2])
    if os.environ.get("DCF_STAMP_RAW"):
        for wg in (0, 1, 8):
            for it in range(3):
                print("wg", wg, "item", it, [int(v) for v in st[wg, it]])
    live = st[:, :, 8] > 0
    nwg = int(live[:, 0].sum())
    # s_memtime counts per XCD (workgroups on different XCDs read different counters): everything below is per workgroup
    first = st[:, 0, 0].astype(np.float64)
    nit = live.sum(axis=1)
    last = np.array([st[w, nit[w] - 1, 8] if nit[w] else 0 for w in range(st.shape[0])], dtype=np.float64)
    span = (last - first)[nit > 0]
    tpu = float(np.median(span)) / max(us - 2.0, 1.0)          # ticks per us (launch + dispatch: ~2 us of a launch's interval)
    print("chain %s x %d layers%s: %.1f us by events = %.2f us per layer; %d workgroups with tiles; median workgroup span %.0f ticks -> %.0f ticks/us"
          % (args.shape, n, " (dgrad)" if args.dgrad else (" (plain)" if args.plain else ""), us, us / n, nwg, np.median(span), tpu))
    d = np.diff(st[:, :, :9], axis=2).astype(np.float64)
    later = live.copy(); later[:, 0] = False                   # items behind the first one (the first has nothing to wait for)
    print("phase                                              median us   p90 us   | first item of a workgroup (median)")
    for k, name in enumerate(PH):
        v = d[:, :, k][later] / tpu
        f = d[:, 0, k][live[:, 0]] / tpu
        print("%-50s %9.2f %8.2f   | %8.2f" % (name, np.median(v), np.percentile(v, 90), np.median(f)))
    gap = (st[:, 1:, 0] - st[:, :-1, 8]).astype(np.float64)[later[:, 1:]] / tpu
    print("%-50s %9.2f %8.2f" % ("between items (next tile's setup)", np.median(gap), np.percentile(gap, 90)))
    per = (st[:, 1:, 8] - st[:, :-1, 8]).astype(np.float64)[later[:, 1:]] / tpu
    print("%-50s %9.2f %8.2f" % ("publish -> next publish (one layer on one CU)", np.median(per), np.percentile(per, 90)))


if __name__ == "__main__":
    main()
