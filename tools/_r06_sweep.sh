mkdir -p gpurun_out/r06a
python bench.py --no-cpu-baseline > gpurun_out/r06a/bench_sweep.json 2>gpurun_out/r06a/bench_sweep.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06a/bench_sweep.json"))
print(d["value"], d["config"]["batch_sweep_frames_per_s"], d["config"]["batch_sweep_ms_per_step"])
PY
tail -3 gpurun_out/r06a/bench_sweep.err
