cd "$GRAFT_REPO_ROOT"
CFG4="--dtype f16 --batch 4 --points 120000 --knn 5 --image-stream resnet50"
run() {
  python3 bench.py --no-cpu-baseline --no-from-host $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k['kernel']:k['ms_per_step'] for k in d['kernel_breakdown']}
print('$1', d['value'], d['ms_per_step'], {k:v for k,v in kb.items() if 'wgrad1s' in k or 'wgrad_grp' in k or k.startswith('conv_wgrad_')})"
}
for r in 1 2; do
run cfg2_all ""; DCF_WGRAD1S_MIN_CH=128 run cfg2_min128 ""; DCF_WGRAD1S=0 run cfg2_generic ""
done
run cfg4_all "$CFG4"; DCF_WGRAD1S_MIN_CH=128 run cfg4_min128 "$CFG4"
run cfg4_all "$CFG4"; DCF_WGRAD1S_MIN_CH=128 run cfg4_min128 "$CFG4"
