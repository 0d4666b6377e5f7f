cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "shared_staging" 2>&1 | tail -3
run() {
  python3 bench.py --no-cpu-baseline --no-from-host --no-roofline $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
}
CFG5="--batch 1 --points 300000 --image 1920x1080"
for r in 1 2; do
run cfg5_wg1 "$CFG5"; DCF_WGRAD1S=0 run cfg5_generic "$CFG5"
run b1_wg1 "--batch 1"; DCF_WGRAD1S=0 run b1_generic "--batch 1"
run cfg2_wg1 ""; DCF_WGRAD1S=0 run cfg2_generic ""
run bn_wg1 "--bn-mode train"; DCF_WGRAD1S=0 run bn_generic "--bn-mode train"
done
