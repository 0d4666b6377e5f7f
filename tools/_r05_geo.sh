#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_geometry.py tests/test_data_path.py tests/test_gpu_benchsize.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3 4; do
python3 bench.py --no-cpu-baseline --no-roofline > gpurun_out/r05i_bench_default_$i.json 2>> gpurun_out/r05i.err
done
python3 bench.py --no-cpu-baseline --batch 8 --no-roofline > gpurun_out/r05i_bench_b8.json 2>> gpurun_out/r05i.err
python3 bench.py --no-cpu-baseline --batch 1 --no-roofline > gpurun_out/r05i_bench_b1.json 2>> gpurun_out/r05i.err
for f in gpurun_out/r05i_bench_*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['config'].get('resident_frames_per_s'), d.get('slowest_step_index'), d['ms_per_step_min_max'])"; done
tail -3 gpurun_out/r05i.err
