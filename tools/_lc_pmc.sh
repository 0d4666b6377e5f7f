cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "6 50" "9 32" "4 64"; do set -- $cfg; export DCF_LC_TH=$1 DCF_LC_TW=$2; 
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d gpurun_out/lcpmc_$1x$2 -- python3 tools/rw_time.py --std --batch 2 l3 > /dev/null 2>&1
python3 tools/sq_summary.py gpurun_out/lcpmc_$1x$2.csv gpurun_out/lcpmc_$1x$2 --match k_conv3x3_lc > /dev/null; rm -rf gpurun_out/lcpmc_$1x$2; echo "== TH x TW = $1 x $2"; python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/lcpmc_$1x$2.csv')):
    print({k:v for k,v in r.items() if k!='kernel'})
PY
done
