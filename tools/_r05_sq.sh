#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/sq_profile.sh r05k "" 2>&1 | tail -3
bash tools/sq_profile.sh r05k "--chain" _chain 2>&1 | tail -3
ls -la gpurun_out/r05k*_sq.csv
