#!/bin/bash
# train-mode BatchNorm: tests, then the bench line with graphs (auto) and eager
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${TAG:-r05n}
timeout 900 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_elementwise.py -m gpu -x -q -k "batchnorm or bn or graph or train_mode" > gpurun_out/${T}_pytest_bn.log 2>&1
tail -5 gpurun_out/${T}_pytest_bn.log
python3 bench.py --bn-mode train --no-cpu-baseline > gpurun_out/${T}_trainbn_bench_line.json 2> gpurun_out/${T}_trainbn.err
python3 bench.py --bn-mode train --graphs off --no-cpu-baseline > gpurun_out/${T}_trainbn_eager_bench_line.json 2> gpurun_out/${T}_trainbn_eager.err
for f in gpurun_out/${T}_trainbn_bench_line.json gpurun_out/${T}_trainbn_eager_bench_line.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config'].get('resident_frames_per_s'))
for k in d.get('kernel_breakdown',[])[:14]: print('   ', k['kernel'], k['ms_per_step'], k['calls_per_step'])"; done
tail -3 gpurun_out/${T}_trainbn.err
