cd $GRAFT_REPO_ROOT
P=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
for v in $VARIANTS; do for cfg in "6 50" "9 32" "4 64"; do set -- $cfg; echo -n "$v "; DCF_LC_KIND=0 DCF_LC_TH=$1 DCF_LC_TW=$2 DCF_HIP_LIB=$PWD/$P/libdcf_hip_v$v.so python3 tools/lc_stamps.py --summary l3 2>&1 | grep "TH/TW"; done; done
