#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_geometry.py -m gpu -x -q -k project_filter_batch 2>&1 | grep -E "assert|Error|equal|^E" | head -20
REPS=2 python3 tools/fromhost_hiccup.py 2>&1 | grep -v amdgpu.ids

