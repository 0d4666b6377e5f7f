P=$PWD/deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
for v in "" _vabl1 _vabl2 _vabl3 _vabl8 _vabl11; do
  DCF_HIP_LIB=$P/libdcf_hip$v.so python tools/rs_time.py 2x44x50x256 2x24x78x256 2x12x39x512 2x88x100x192 2>&1 | tail -1
done
