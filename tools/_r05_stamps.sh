#!/bin/bash
cd $GRAFT_REPO_ROOT
P=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
mkdir -p gpurun_out
python3 tools/chain_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05e_chain_time.txt
export DCF_HIP_LIB=$PWD/$P/libdcf_hip_vstamp.so
for s in "2x44x50x256 11" "2x88x100x192 11" "2x176x200x128 7" "2x94x311x64 4"; do
  python3 tools/chain_stamps.py $s 2>&1 | grep -v amdgpu.ids
  python3 tools/chain_stamps.py $s --dgrad 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05e_chain_stamps.txt 2>&1
cat gpurun_out/r05e_chain_stamps.txt
