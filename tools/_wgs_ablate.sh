#!/bin/bash
# timing ablations of the shared-staging weight-gradient kernel: rebuilds conv_wgs.o on the GPU box per mask
cd "$(dirname "$0")/.." || exit 1
CS=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd/csrc
for m in "$@"; do
    touch $CS/conv_wgs.hip
    make -s -C $CS "CXXFLAGS=--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DDCF_WGS_DBG_MASK=$m" > /dev/null 2>&1 || { echo build failed; exit 1; }
    echo "== mask $m"
    python tools/wgs_bench.py l3 conv3
done
