export DCF_HIP_LIB=$PWD/deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd/libdcf_hip_vnl4.so
python tools/rs_pf_ab.py --opt RS_NL4 l4 i1 i2 l3 2>&1 | tail -8
python tools/rs_pf_ab.py --opt RS_NL4 --batch 1 l4 i1 i2 l3 2>&1 | tail -8
