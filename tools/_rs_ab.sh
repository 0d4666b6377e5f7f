#!/bin/bash
# A/B of a compile-time switch of conv_rs.hip on the GPU box: bash tools/_rs_ab.sh "-DDCF_RS_SPREAD=0" "-DDCF_RS_SPREAD=1"
cd "$(dirname "$0")/.." || exit 1
CS=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd/csrc
for f in "$@"; do
    touch $CS/conv_rs.hip
    make -s -C $CS "CXXFLAGS=--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $f" > /dev/null 2>&1 || { echo build failed; exit 1; }
    echo "== $f"
    python tools/conv_bench.py 2>&1 | grep -E "^(l3|l4|l5|conv3|l2|i1|i2|i3|i4) |totals"
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-from-host --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
