cd "$GRAFT_REPO_ROOT"
# A/B of the consumer + loader wave forms of the row-sharing kernel (RS_L16: small-M kind, 8 + 8 waves; RS_L12: kinds 1 / 0, 8 + 4 waves)
DCF_RS_L12=3 timeout 900 python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "conv_fwd or conv_dgrad or sixteen" 2>&1 | tail -3
for m in 0 1 3; do echo "=== RS_L12=$m"; DCF_RS_L12=$m python3 tools/conv_bench.py 2>&1 | grep -E "^l2 |^l3 |^l4 |^conv3|^i1 |^i2 "; done
run() {
  python3 bench.py --no-cpu-baseline --no-from-host --no-roofline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
}
for r in 1 2; do
DCF_RS_L12=0 run l12_0; DCF_RS_L12=1 run l12_1; DCF_RS_L12=3 run l12_3
done
