# the final tree of round 6: full profile set, GPU suite, three stress runs (tools/_gpurun_retry.sh 3400 'bash tools/_r06_final.sh r06e')
T=${1:-r06e}
bash tools/_prof_all.sh $T > gpurun_out/prof_all_$T.log 2>&1
tail -12 gpurun_out/prof_all_$T.log
python -m pytest tests -m gpu -q -rA > gpurun_out/${T}_pytest_gpu.log 2>&1
tail -3 gpurun_out/${T}_pytest_gpu.log
mkdir -p gpurun_out/$T
for i in 1 2 3; do
  python -m pytest tests/test_gpu_stress.py -m gpu -q -rA > gpurun_out/$T/stress_run$i.log 2>&1
  grep -E "^stress (stream|sibling):|passed|failed" gpurun_out/$T/stress_run$i.log | cut -c1-300
done
