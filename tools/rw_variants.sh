#!/bin/bash
# Compile-time variants of one kernel file (default conv_rw.hip; KFILE=conv_lc picks another), each linked with the product's
# other objects into <pkg>/libdcf_hip_v<name>.so (they travel with gpurun; DCF_HIP_LIB selects one):
#   [KFILE=conv_lc] bash tools/rw_variants.sh name1="-DRW_DBG=5" name2="-DFOO" ...
# Run `make -C <pkg>/csrc` first: the other objects are taken from there.
cd "$(dirname "$0")/.." || exit 1
P=deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd
CS=$P/csrc
mkdir -p /tmp/rwv
KFILE=${KFILE:-conv_rw}
OTHERS=$(ls $CS/*.o | grep -v $KFILE.o)
pids=()
for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-pass-failed -DRW_BF16_ONLY -DLC_BF16_ONLY $flags -c $CS/$KFILE.hip -o /tmp/rwv/${KFILE}_$name.o 2> /tmp/rwv/$name.err \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libdcf_hip_v$name.so $OTHERS /tmp/rwv/${KFILE}_$name.o && echo "built $name" || { echo "FAILED $name"; tail -5 /tmp/rwv/$name.err; } ) &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
