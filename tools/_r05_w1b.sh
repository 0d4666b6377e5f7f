cd "$GRAFT_REPO_ROOT"
CFG4="--dtype f16 --batch 4 --points 120000 --knn 5 --image-stream resnet50"
run() {
  python3 bench.py --no-cpu-baseline --no-from-host $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k['kernel']:k['ms_per_step'] for k in d['kernel_breakdown']}
print('$1', d['value'], d['ms_per_step'], {k:v for k,v in kb.items() if 'wgrad' in k})"
}
echo "== cfg4"
run wg1 "$CFG4"
DCF_WGRAD1S=0 run generic "$CFG4"
DCF_WGRAD1S_GRANULE=128 run wg1_full_tiles "$CFG4"
DCF_WGRAD1S_STAGES=3 run wg1_ns3 "$CFG4"
DCF_WGRAD1S_BLOCKS=288 run wg1_blocks288 "$CFG4"
run wg1 "$CFG4"
DCF_WGRAD1S=0 run generic "$CFG4"
echo "== cfg2"
DCF_WGRAD1S_GRANULE=128 run wg1_full_tiles ""
DCF_WGRAD1S=0 run generic ""
