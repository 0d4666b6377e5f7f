#!/bin/bash
# round 5, first GPU call: data-path + model tests, then the default bench line (resident + from-host legs) twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_data_path.py tests/test_gpu_model.py tests/test_cabi.py -m gpu -x -q > gpurun_out/r05a_pytest.log 2>&1
tail -3 gpurun_out/r05a_pytest.log
python3 bench.py --no-cpu-baseline > gpurun_out/r05a_bench_line.json 2> gpurun_out/r05a_bench.err
python3 bench.py --no-cpu-baseline --no-roofline > gpurun_out/r05a_bench_line2.json 2>> gpurun_out/r05a_bench.err
python3 bench.py --no-cpu-baseline --no-roofline --from-host > gpurun_out/r05a_bench_line_fh.json 2>> gpurun_out/r05a_bench.err
python3 bench.py --no-cpu-baseline --no-roofline --batch 1 > gpurun_out/r05a_bench_line_b1.json 2>> gpurun_out/r05a_bench.err
for f in gpurun_out/r05a_bench_line.json gpurun_out/r05a_bench_line2.json gpurun_out/r05a_bench_line_fh.json gpurun_out/r05a_bench_line_b1.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['config']['input'], d['config'].get('from_host_frames_per_s'), (d.get('roofline') or {}).get('kernel'))"; done
tail -5 gpurun_out/r05a_bench.err
