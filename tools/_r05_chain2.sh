#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_model.py -m gpu -x -q -k "chain or cfg2_size or tiny or hip_graph or train_step" > gpurun_out/r05c_model_pytest.log 2>&1
tail -8 gpurun_out/r05c_model_pytest.log
for i in 1 2; do
DCF_CHAIN=0 python3 bench.py --no-cpu-baseline --no-roofline > gpurun_out/r05c_bench_nochain_$i.json 2>> gpurun_out/r05c_bench.err
DCF_CHAIN=1 python3 bench.py --no-cpu-baseline --no-roofline > gpurun_out/r05c_bench_chain_$i.json 2>> gpurun_out/r05c_bench.err
done
python3 bench.py --no-cpu-baseline > gpurun_out/r05c_bench_chain_roof.json 2>> gpurun_out/r05c_bench.err
for f in gpurun_out/r05c_bench_*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['config'].get('from_host_frames_per_s'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))"; done
tail -5 gpurun_out/r05c_bench.err
