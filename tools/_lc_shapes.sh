cd $GRAFT_REPO_ROOT
P=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
for cfg in "6 50" "6 48" "9 32" "4 64" "8 40" "10 30" "3 100" "5 56" "7 44" "2 128"; do set -- $cfg; DCF_LC_KIND=0 DCF_LC_TH=$1 DCF_LC_TW=$2 DCF_HIP_LIB=$PWD/$P/libdcf_hip_vstamp.so python3 tools/lc_stamps.py --summary ${3:-l3} 2>&1 | grep -v amdgpu; done
