#!/bin/bash
# Time every variant library built by tools/rw_variants.sh (GPU box): bash tools/rw_vrun.sh "--batch 8 l3 l4" [name ...]
cd "$(dirname "$0")/.." || exit 1
P=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
ARGS=$1; shift
NAMES="$@"
if [ -z "$NAMES" ]; then NAMES=$(ls $P/libdcf_hip_v*.so | sed 's/.*libdcf_hip_v\(.*\)\.so/\1/'); fi
for n in $NAMES; do DCF_HIP_LIB=$PWD/$P/libdcf_hip_v$n.so python3 tools/rw_time.py --tag $n $ARGS 2>&1 | grep -v amdgpu.ids; done
