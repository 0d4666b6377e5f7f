#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do HB=8 REPS=1 python3 tools/fromhost_hiccup.py 2>&1 | grep -v amdgpu.ids | grep -v "worker +"; done
for i in 1 2; do HB=2 REPS=1 python3 tools/fromhost_hiccup.py 2>&1 | grep -v amdgpu.ids | grep -v "worker +"; done
