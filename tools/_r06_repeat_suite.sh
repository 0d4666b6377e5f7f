mkdir -p gpurun_out/r06d
for i in 1 2; do
  python -m pytest tests -m gpu -q > gpurun_out/r06d/pytest_gpu_run$i.log 2>&1
  tail -2 gpurun_out/r06d/pytest_gpu_run$i.log
done
python bench.py > gpurun_out/r06d/bench_line.json 2> gpurun_out/r06d/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06d/bench_line.json"))
print(d["value"], d["ms_per_step"], d["config"]["resident_frames_per_s"], d["config"]["batch_sweep_frames_per_s"], d["roofline"]["kernel"][:50], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_stale"])
PY
