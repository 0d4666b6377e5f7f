cd $GRAFT_REPO_ROOT
T=${1:-r03b}
bash tools/profile_round.sh $T "" > gpurun_out/prof_$T.log 2>&1
for b in 1 4 8; do python3 bench.py --batch $b --no-cpu-baseline --no-batch-sweep > gpurun_out/${T}_cfg2_b${b}_bench_line.json 2> gpurun_out/${T}_b$b.err; done
bash tools/profile_round.sh $T "--dtype f16 --batch 4 --points 120000 --knn 5 --image-stream resnet50" _cfg4 > gpurun_out/prof_${T}_cfg4.log 2>&1
python3 bench.py --batch 1 --points 300000 --image 1920x1080 --no-cpu-baseline --no-batch-sweep > gpurun_out/${T}_cfg5shape_bf16_bench_line.json 2> gpurun_out/${T}_cfg5a.err
python3 bench.py --batch 1 --points 300000 --image 1920x1080 --dtype fp8 --no-cpu-baseline --no-batch-sweep > gpurun_out/${T}_cfg5_fp8_bench_line.json 2> gpurun_out/${T}_cfg5b.err
python3 bench.py --loss-sampling device --no-cpu-baseline --no-batch-sweep > gpurun_out/${T}_cfg2_device_sampling_bench_line.json 2> gpurun_out/${T}_dev.err
# round 4: cfg5 (fp8 forward) with its own kernel stats and PMC passes; the train-mode BN line; batch 1 with and without graphs
bash tools/profile_round.sh $T "--batch 1 --points 300000 --image 1920x1080 --dtype fp8" _cfg5_fp8 > gpurun_out/prof_${T}_cfg5_fp8.log 2>&1
python3 bench.py --bn-mode train --no-cpu-baseline --no-batch-sweep > gpurun_out/${T}_trainbn_bench_line.json 2> gpurun_out/${T}_trainbn.err
python3 bench.py --batch 1 --graphs off --no-cpu-baseline --no-batch-sweep > gpurun_out/${T}_cfg2_b1_eager_bench_line.json 2> gpurun_out/${T}_b1e.err
for f in gpurun_out/prof_$T/bench_line.json gpurun_out/${T}_cfg2_b1_bench_line.json gpurun_out/${T}_cfg2_b4_bench_line.json gpurun_out/${T}_cfg2_b8_bench_line.json gpurun_out/prof_${T}_cfg4/bench_line.json gpurun_out/${T}_cfg5shape_bf16_bench_line.json gpurun_out/${T}_cfg5_fp8_bench_line.json gpurun_out/${T}_cfg2_device_sampling_bench_line.json gpurun_out/prof_${T}_cfg5_fp8/bench_line.json gpurun_out/${T}_trainbn_bench_line.json gpurun_out/${T}_cfg2_b1_eager_bench_line.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], (d.get('from_host') or {}).get('value'))
except Exception as e: print('$f', 'ERR', e)"; done
tail -3 gpurun_out/prof_$T.log
