#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG:-r05b}_pytest_gpu.log 2>&1
tail -5 gpurun_out/${TAG:-r05b}_pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/_prof_all.sh ${TAG:-r05b} 2>&1 | tail -20
python3 bench.py --chain --no-cpu-baseline > gpurun_out/${TAG:-r05b}_cfg2_chain_bench_line.json 2> gpurun_out/${TAG:-r05b}_chain.err
python3 bench.py --chain --batch 1 --no-cpu-baseline --no-roofline > gpurun_out/${TAG:-r05b}_cfg2_b1_chain_bench_line.json 2>> gpurun_out/${TAG:-r05b}_chain.err
for f in gpurun_out/${TAG:-r05b}_cfg2_chain_bench_line.json gpurun_out/${TAG:-r05b}_cfg2_b1_chain_bench_line.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config'].get('resident_frames_per_s'), d['config'].get('conv_chain'))"; done
