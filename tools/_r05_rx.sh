cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "wgrad" 2>&1 | tail -2
CFG4="--dtype f16 --batch 4 --points 120000 --knn 5 --image-stream resnet50"
run() {
  python3 bench.py --no-cpu-baseline --no-from-host $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k['kernel']:k['ms_per_step'] for k in d['kernel_breakdown']}
print('$1', d['value'], d['ms_per_step'], {k:v for k,v in kb.items() if 'wgrad' in k and 'grp' in k})"
}
for r in 1 2; do
run cfg2_new ""; DCF_WGRAD_RANGE_XCD=0 run cfg2_old ""
done
run cfg4_new "$CFG4"; DCF_WGRAD_RANGE_XCD=0 run cfg4_old "$CFG4"
TAG=r05v bash tools/_r05_wt.sh | tail -30
