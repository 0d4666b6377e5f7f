#!/bin/bash
# A/B of a compile-time switch of conv_rs.hip on the GPU box: bash tools/_rs_ab.sh "-DDCF_RS_SPREAD=0" "-DDCF_RS_SPREAD=1"
cd "$(dirname "$0")/.." || exit 1
CS=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd/csrc
for f in "$@"; do
    touch $CS/conv_rs.hip
    make -s -C $CS "CXXFLAGS=--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $f" > /dev/null 2>&1 || { echo build failed; exit 1; }
    echo "== $f"
    for r in 1 2 3; do python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-from-host --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
    python bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-from-host --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b8', d['value'], d['ms_per_step'])"
done
