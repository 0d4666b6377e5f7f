mkdir -p gpurun_out/r06c
python -m pytest tests -m gpu -q -rA > gpurun_out/r06c_pytest_gpu.log 2>&1
tail -4 gpurun_out/r06c_pytest_gpu.log
for i in 1 2 3; do
  DCF_NO_DP_CHILDREN= python -m pytest tests/test_gpu_stress.py -m gpu -q -rA > gpurun_out/r06c/stress_run$i.log 2>&1
  grep -E "^stress (stream|sibling):|passed|failed" gpurun_out/r06c/stress_run$i.log | cut -c1-400
done
