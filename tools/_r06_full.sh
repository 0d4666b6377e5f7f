mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q -rA 2>&1 | tail -60 > gpurun_out/r06a/pytest_gpu.log
tail -5 gpurun_out/r06a/pytest_gpu.log
python bench.py > gpurun_out/r06a/bench_line.json 2> gpurun_out/r06a/bench_err.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06a/bench_line.json'))
print(d['value'], d['ms_per_step'], d['config']['resident_frames_per_s'], d['roofline']['kernel'][:60], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for c in d['kernel_classes']: print(c)
PY
