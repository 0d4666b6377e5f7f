# kernels of a cfg2 step with their grid sizes and durations (small grids with long durations = serial work on the critical path)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gt && mkdir -p gpurun_out/gt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-from-host > /dev/null 2> gpurun_out/gt/err.txt
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/gt/*/*kernel_trace.csv')[0]
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0, set()])
for r in csv.DictReader(open(f)):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:70]
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(wg, 1)
    a = agg[(name, grid)]
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; a[4].add(r.get('Stream_Id', r.get('Queue_Id', '')))
rows = sorted(agg.items(), key=lambda kv: -kv[1][1] / kv[1][0])
print("%-72s %8s %6s %9s  queue" % ("kernel", "wgs", "calls", "avg us"))
for (name, grid), a in rows:
    if grid <= 64 and a[1] / a[0] >= 6.0:
        print("%-72s %8d %6d %9.1f  %s" % (name, grid, a[0], a[1] / a[0], ",".join(sorted(a[4]))))
PY
rm -rf gpurun_out/gt
