cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02j "" > gpurun_out/prof_r02j.log 2>&1
for b in 1 4 8; do python3 bench.py --batch $b --no-cpu-baseline > gpurun_out/r02j_cfg2_b${b}_bench_line.json 2> gpurun_out/r02j_b$b.err; done
bash tools/profile_round.sh r02j "--dtype f16 --batch 4 --points 120000 --knn 5 --image-stream resnet50" _cfg4 > gpurun_out/prof_r02j_cfg4.log 2>&1
python3 bench.py --batch 1 --points 300000 --image 1920x1080 --no-cpu-baseline > gpurun_out/r02j_cfg5shape_bf16_bench_line.json 2> gpurun_out/r02j_cfg5a.err
python3 bench.py --batch 1 --points 300000 --image 1920x1080 --dtype fp8 --no-cpu-baseline > gpurun_out/r02j_cfg5_fp8_bench_line.json 2> gpurun_out/r02j_cfg5b.err
for f in gpurun_out/prof_r02j/bench_line.json gpurun_out/r02j_cfg2_b1_bench_line.json gpurun_out/r02j_cfg2_b4_bench_line.json gpurun_out/r02j_cfg2_b8_bench_line.json gpurun_out/prof_r02j_cfg4/bench_line.json gpurun_out/r02j_cfg5shape_bf16_bench_line.json gpurun_out/r02j_cfg5_fp8_bench_line.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
except Exception as e: print('$f', 'ERR', e)"; done
tail -3 gpurun_out/prof_r02j.log
