cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in off on; do
rm -rf gpurun_out/gt && mkdir -p gpurun_out/gt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gt -- python3 bench.py --bn-mode train --graphs $mode --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --input resident --no-other-leg > gpurun_out/gt/line.json 2> gpurun_out/gt/err.txt
echo "=== train-mode BN, batch 2, graphs $mode"; python3 -c "
import json; d=json.loads(open('gpurun_out/gt/line.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
python3 tools/timeline.py $(ls gpurun_out/gt/*/*kernel_trace.csv | head -1) 2>&1 | head -30
done > gpurun_out/r05n_timeline_trainbn.txt 2>&1
rm -rf gpurun_out/gt
cat gpurun_out/r05n_timeline_trainbn.txt
