#!/bin/bash
# Compile-time variants of one kernel file, each linked with the product's other objects into <pkg>/libdcf_hip_v<name>.so (they
# travel with gpurun; DCF_HIP_LIB selects one).  Default KFILE=conv_rw = the experimental register-weight convolution of
# tools/variants/ (not part of libdcf_hip.so: the variant library ADDS its four entry points, tools/variants/conv_rw.h);
# KFILE=conv_lc / conv_rs / geometry ... rebuilds a product kernel file with extra flags in place of the product's object:
#   [KFILE=conv_lc] bash tools/rw_variants.sh name1="-DRW_DBG=5" name2="-DFOO" ...
# Run `make -C <pkg>/csrc` first: the other objects are taken from there.
cd "$(dirname "$0")/.." || exit 1
P=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
CS=$P/csrc
mkdir -p /tmp/rwv
KFILE=${KFILE:-conv_rw}
OTHERS=$(ls $CS/*.o | grep -v "/$KFILE.o")
SRC=$CS/$KFILE.hip
[ -f $SRC ] || SRC=tools/variants/$KFILE.hip
pids=()
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed -DRW_BF16_ONLY -DLC_BF16_ONLY -DRS_BF16_ONLY $flags -I $CS -I tools/variants -c $SRC -o /tmp/rwv/${KFILE}_$name.o 2> /tmp/rwv/$name.err \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libdcf_hip_v$name.so $OTHERS /tmp/rwv/${KFILE}_$name.o && echo "built $name" || { echo "FAILED $name"; tail -5 /tmp/rwv/$name.err; } ) &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
