#!/bin/bash
# SQ counter passes of the benchmark step (GPU box): where the waves of the convolution kernels spend their cycles.
#   bash tools/sq_profile.sh r04a ["extra bench args"] [suffix]  ->  gpurun_out/<tag><suffix>_sq.csv
# Two --pmc passes (8 SQ slots each), no trace flags; the program itself follows `--` (no shell / env hop: pool rule).
set -u
TAG=$1; EXTRA=${2:-}; SUF=${3:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sq_$TAG$SUF
mkdir -p $OUT
B="bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-batch-sweep --no-from-host $EXTRA"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $B > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p2 -- python3 $B > /dev/null 2> $OUT/p2.err
python3 tools/sq_summary.py gpurun_out/${TAG}${SUF}_sq.csv $OUT/p1 $OUT/p2 --match k_conv,k_knn,k_fusion,k_wgrad
tail -2 $OUT/p1.err $OUT/p2.err
rm -rf $OUT/p1 $OUT/p2
