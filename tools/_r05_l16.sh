cd "$GRAFT_REPO_ROOT"
DCF_RS_L16=7 timeout 900 python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "conv_fwd or conv_dgrad" 2>&1 | tail -3
for m in 0 1 3 7; do echo "=== RS_L16=$m"; DCF_RS_L16=$m python3 tools/conv_bench.py 2>&1 | grep -E "^l3 |^l4 |^l5 |^conv3|^i1 |^i2 |^i3 |^i4 "; done
run() {
  python3 bench.py --no-cpu-baseline --no-from-host --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
}
for r in 1 2; do
DCF_RS_L16=0 run l16_0; DCF_RS_L16=1 run l16_1; DCF_RS_L16=3 run l16_3; DCF_RS_L16=7 run l16_7
done
