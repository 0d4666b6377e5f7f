#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_conv_chain.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python3 -m pytest tests/test_gpu_model.py -m gpu -x -q -k "chain" 2>&1 | tail -4
python3 tools/chain_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05g_chain_time.txt
for i in 1 2; do
DCF_CHAIN=0 python3 bench.py --no-cpu-baseline --no-roofline --input resident --no-other-leg > gpurun_out/r05g_bench_nochain_$i.json 2>> gpurun_out/r05g_bench.err
DCF_CHAIN=1 python3 bench.py --no-cpu-baseline --no-roofline --input resident --no-other-leg > gpurun_out/r05g_bench_chain_$i.json 2>> gpurun_out/r05g_bench.err
done
DCF_CHAIN=0 python3 bench.py --no-cpu-baseline --no-roofline --batch 1 --input resident --no-other-leg > gpurun_out/r05g_bench_b1_nochain.json 2>> gpurun_out/r05g_bench.err
DCF_CHAIN=1 python3 bench.py --no-cpu-baseline --no-roofline --batch 1 --input resident --no-other-leg > gpurun_out/r05g_bench_b1_chain.json 2>> gpurun_out/r05g_bench.err
for f in gpurun_out/r05g_bench_*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; done
tail -3 gpurun_out/r05g_bench.err
