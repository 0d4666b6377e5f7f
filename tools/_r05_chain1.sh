#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_conv_chain.py -m gpu -x -q > gpurun_out/r05b_chain_pytest.log 2>&1
tail -15 gpurun_out/r05b_chain_pytest.log
timeout 600 python3 -m pytest tests/test_data_path.py -m gpu -x -q > gpurun_out/r05b_data_pytest.log 2>&1
tail -5 gpurun_out/r05b_data_pytest.log
