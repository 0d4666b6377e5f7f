cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gt && mkdir -p gpurun_out/gt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gt -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-from-host > /dev/null 2> gpurun_out/gt/err.txt
python3 tools/timeline.py $(ls gpurun_out/gt/*/*kernel_trace.csv | head -1) 2>&1 | head -60
rm -rf gpurun_out/gt
