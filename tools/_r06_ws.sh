export DCF_HIP_LIB=$PWD/deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd/libdcf_hip_vwstamp.so
mkdir -p gpurun_out/ws2
python tools/rs_wstamps.py 2x88x100x192 > gpurun_out/ws2/l4_fwd.txt 2>&1
python tools/rs_wstamps.py 2x176x200x128 > gpurun_out/ws2/l3_fwd.txt 2>&1
python tools/rs_wstamps.py 2x176x200x128 --dgrad > gpurun_out/ws2/l3_dgrad.txt 2>&1
python tools/rs_wstamps.py 2x44x50x256 > gpurun_out/ws2/l5_fwd.txt 2>&1
head -12 gpurun_out/ws2/l4_fwd.txt; head -12 gpurun_out/ws2/l3_fwd.txt;  head -12 gpurun_out/ws2/l5_fwd.txt
unset DCF_HIP_LIB
python tools/rs_pf_ab.py l3 l4 l5 i1 i2 2>&1 | tail -7
