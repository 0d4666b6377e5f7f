#!/bin/bash
# SQ counters of tools/rw_bench.py (GPU box): bash tools/rw_sq.sh TAG "rw_bench args"
set -u
TAG=$1; EXTRA=${2:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sqrw_$TAG
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 tools/rw_bench.py $EXTRA > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p2 -- python3 tools/rw_bench.py $EXTRA > /dev/null 2> $OUT/p2.err
python3 tools/sq_summary.py gpurun_out/${TAG}_sqrw.csv $OUT/p1 $OUT/p2 --match k_conv3x3
rm -rf $OUT/p1 $OUT/p2
