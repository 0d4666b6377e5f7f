#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one or more `rocprofv3 --pmc ...` passes (counter_collection.csv files).

    python3 tools/sq_summary.py OUT.csv PASS_DIR [PASS_DIR ...] [--match substr[,substr...]]

One row per GPU function (the kernel name up to its argument list), one column per counter: the mean counter value per
dispatch, plus the number of dispatches seen.  The SQ_* wave counters of gfx950 count quad-cycles summed over the waves
(MI355X_MICROARCH.md, "rocprofv3 PMC slots"): WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES, so the derived
columns are shares of SQ_WAVE_CYCLES: parked at a wait / barrier, stalled at issue (of which: on the LDS), issuing."""
import collections
import csv
import glob
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--match")]
    match = None
    for i, a in enumerate(sys.argv):
        if a == "--match":
            match = sys.argv[i + 1].split(",")
            args.remove(sys.argv[i + 1])
    out, dirs = args[0], args[1:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    counters = []
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if match and not any(m in k for m in match):
                    continue
                c = r["Counter_Name"]
                if c not in counters:
                    counters.append(c)
                acc[k][c] += float(r["Counter_Value"])
                cnt[k][c].add((f, r["Dispatch_Id"]))
    rows = []
    for k in acc:
        row = {"kernel": k, "dispatches": max(len(s) for s in cnt[k].values())}
        for c in counters:
            n = len(cnt[k][c])
            row[c] = acc[k][c] / n if n else ""
        wc = row.get("SQ_WAVE_CYCLES") or 0
        if wc:
            for name, c in (("share_wait_any", "SQ_WAIT_ANY"), ("share_wait_inst_any", "SQ_WAIT_INST_ANY"),
                            ("share_wait_inst_lds", "SQ_WAIT_INST_LDS"), ("share_active_inst_any", "SQ_ACTIVE_INST_ANY")):
                if row.get(c) not in ("", None):
                    row[name] = round(row[c] / wc, 4)
        rows.append(row)
    rows.sort(key=lambda r: -(r.get("SQ_WAVE_CYCLES") or 0) * r["dispatches"])
    cols = ["kernel", "dispatches"] + counters + ["share_wait_any", "share_wait_inst_any", "share_wait_inst_lds", "share_active_inst_any"]
    with open(out, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=cols, extrasaction="ignore")
        w.writeheader()
        for r in rows:
            w.writerow({c: (("%.6g" % r[c]) if isinstance(r.get(c), float) else r.get(c, "")) for c in cols})
    print("wrote", out, len(rows), "kernels")


if __name__ == "__main__":
    main()
