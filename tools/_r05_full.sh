#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG:-r05b}_pytest_gpu.log 2>&1
tail -5 gpurun_out/${TAG:-r05b}_pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/_prof_all.sh ${TAG:-r05b} 2>&1 | tail -20
