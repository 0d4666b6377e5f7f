P=$PWD/deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
for i in 1 2; do
for v in "" _vprio1 _vprio2; do
  DCF_HIP_LIB=$P/libdcf_hip$v.so python tools/rs_time.py 2x88x100x192 2x176x200x128 2x47x156x128 2x44x50x256 2x176x200x192 2>&1 | tail -1
done
done
