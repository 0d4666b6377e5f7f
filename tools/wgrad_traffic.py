#!/usr/bin/env python3
"""HBM traffic of the weight-gradient kernels, layer by layer (cfg2 shapes, one launch each, the kernels the grouped launches
run): which layers fetch more than their tensors.

  run (GPU box, two passes -- counters are never combined with traces):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/wt_f -- python3 tools/wgrad_traffic.py run
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/wt_w -- python3 tools/wgrad_traffic.py run
  summarise:
    python3 tools/wgrad_traffic.py report gpurun_out/wt_f gpurun_out/wt_w
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (bench.py does the same)."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (name, H, W, Cin, Cout, k, stride, layers of that shape per cfg2 step)
LAYERS = [("l1 3x3", 704, 800, 32, 32, 3, 1, 2), ("l2s 3x3/2", 704, 800, 32, 64, 3, 2, 1), ("l2d 1x1/2", 704, 800, 32, 64, 1, 2, 1),
          ("l2 3x3", 352, 400, 64, 64, 3, 1, 3), ("l3s 3x3/2", 352, 400, 64, 128, 3, 2, 1), ("l3d 1x1/2", 352, 400, 64, 128, 1, 2, 1),
          ("l3 3x3", 176, 200, 128, 128, 3, 1, 7), ("l4s 3x3/2", 176, 200, 128, 192, 3, 2, 1), ("l4d 1x1/2", 176, 200, 128, 192, 1, 2, 1),
          ("l4 3x3", 88, 100, 192, 192, 3, 1, 11), ("l5s 3x3/2", 88, 100, 192, 256, 3, 2, 1), ("l5d 1x1/2", 88, 100, 192, 256, 1, 2, 1),
          ("l5 3x3", 44, 50, 256, 256, 3, 1, 11), ("conv3 3x3", 176, 200, 192, 192, 3, 1, 1), ("lat2 1x1", 176, 200, 128, 192, 1, 1, 1),
          ("lat3 1x1", 88, 100, 192, 192, 1, 1, 1), ("lat5 1x1", 44, 50, 256, 192, 1, 1, 1), ("fc2 site1 1x1", 352, 400, 64, 64, 1, 1, 1),
          ("fc2 site2 1x1", 176, 200, 128, 128, 1, 1, 1), ("i1 3x3", 94, 311, 64, 64, 3, 1, 4), ("i2s 3x3/2", 94, 311, 64, 128, 3, 2, 1),
          ("i2 3x3", 47, 156, 128, 128, 3, 1, 3), ("i3s 3x3/2", 47, 156, 128, 256, 3, 2, 1), ("i3 3x3", 24, 78, 256, 256, 3, 1, 3),
          ("i4s 3x3/2", 24, 78, 256, 512, 3, 2, 1), ("i4 3x3", 12, 39, 512, 512, 3, 1, 3)]
REPS = 3
B = 2


def run():
    import torch
    import bench
    ops = bench.pkg("ops")
    for name, Hh, W, Cin, Cout, k, s, cnt in LAYERS:
        pad = k // 2
        Ho, Wo = (Hh + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = torch.randn(B, Hh, W, Cin, device="cuda").to(torch.bfloat16)
        gy = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
        ns = ops.conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, k, k, s)
        slabs = torch.empty((ns, Cout, k, k, Cin), device="cuda")
        gsum = torch.empty((4 * ns, Cout), device="cuda")
        for _ in range(REPS):
            ops.conv2d_wgrad(1, x, gy, slabs, ns, k, k, s, pad, gsum)
        torch.cuda.synchronize()


def load(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and "wgrad" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def report(fd, wd):
    fr, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    assert len(fr) == len(wr) == len(LAYERS) * REPS, (len(fr), len(wr), len(LAYERS) * REPS)
    print("%-16s %-28s %9s %9s %9s | %9s %9s %6s   x layers" % ("layer", "kernel", "fetch MB", "write MB", "total MB", "tensors MB", "slabs MB", "ratio"))
    tot_m = tot_a = 0.0
    for i, (name, Hh, W, Cin, Cout, k, s, cnt) in enumerate(LAYERS):
        pad = k // 2
        Ho, Wo = (Hh + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        f = 2.0 * fr[i * REPS + REPS - 1][2] / 1e3          # KB -> MB, gfx950 FETCH_SIZE correction
        w = wr[i * REPS + REPS - 1][2] / 1e3
        import re
        m = re.search(r"(k_\w+<[^>]*>|k_\w+)", fr[i * REPS][1])
        kern = (m.group(1) if m else fr[i * REPS][1]).replace("unsigned short", "bf16")[:28]
        tens = B * (Hh * W * Cin + Ho * Wo * Cout) * 2 / 1e6
        print("%-16s %-28s %9.1f %9.1f %9.1f | %9.1f %9.1f %6.2f   x%d" % (name, kern, f, w, f + w, tens, w, (f + w) / (tens + w), cnt))
        tot_m += (f + w) * cnt; tot_a += (tens + w) * cnt
    print("per step (shape counts applied): measured %.0f MB, tensors + slabs %.0f MB, ratio %.2f" % (tot_m, tot_a, tot_m / tot_a))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2], sys.argv[3])
