#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_geometry.py tests/test_data_path.py tests/test_gpu_benchsize.py -m gpu -x -q 2>&1 | tail -5
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do
DCF_VOXEL_HASH=0 python3 bench.py --no-cpu-baseline --no-roofline --input resident --no-other-leg > gpurun_out/r05h_bench_dense_$i.json 2>> gpurun_out/r05h.err
python3 bench.py --no-cpu-baseline --no-roofline --input resident --no-other-leg > gpurun_out/r05h_bench_hash_$i.json 2>> gpurun_out/r05h.err
done
python3 bench.py --no-cpu-baseline > gpurun_out/r05h_bench_default.json 2>> gpurun_out/r05h.err
for f in gpurun_out/r05h_bench_*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['config'].get('resident_frames_per_s'), d.get('slowest_step_index'), d['ms_per_step_min_max'])"; done
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05h_bench_default.json').read().strip().splitlines()[-1])
for b in d['kernel_breakdown']:
    if b['kernel'].startswith(('voxel','project','compact','inv_','knn','scan')): print(b['kernel'], b['ms_per_step'], b['calls_per_step'])
for c in d['kernel_classes']: print(c['class'][:40], c['ms_per_step'])
PY
tail -3 gpurun_out/r05h.err
