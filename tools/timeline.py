#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 --kernel-trace CSV: busy time and idle gaps per HIP queue between consecutive
k_adam launches (one train step).  usage: timeline.py <kernel_trace.csv> [step index]"""
import csv, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) - 3
lo, hi = adam[k], adam[k + 1]
step = rows[lo + 1:hi + 1]
t0, t1 = rows[lo]["e"], rows[hi]["e"]
print("step %d: %.3f ms wall, %d kernels" % (k, (t1 - t0) / 1e6, len(step)))
byq = defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r["e"] - r["s"] for r in rs)
    gaps = []
    prev = None
    pname = ""
    for r in rs:
        if prev is not None and r["s"] > prev:
            gaps.append((r["s"] - prev, "%s  (after %s, %.2f ms into the step)" % (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48],
                                                                                 pname.replace("(anonymous namespace)::", "")[:40], (r["s"] - t0) / 1e6)))
        if prev is None or r["e"] >= prev:
            pname = r["Kernel_Name"]
        prev = max(prev or 0, r["e"])
    print("queue %s: %d kernels, busy %.3f ms, span %.3f ms, idle inside span %.3f ms" % (
        q, len(rs), busy / 1e6, (rs[-1]["e"] - rs[0]["s"]) / 1e6, sum(g for g, _ in gaps) / 1e6))
    gaps.sort(reverse=True)
    for g, n in gaps[:12]:
        print("     gap %.1f us before %s" % (g / 1e3, n))
# union busy over all queues
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
depth, last, busy = 0, None, 0
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
print("GPU busy (any queue) %.3f ms of %.3f ms" % (busy / 1e6, (t1 - t0) / 1e6))
