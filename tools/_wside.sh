run() { python bench.py --no-cpu-baseline --no-from-host --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s' % '$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['config'].get('final_loss'))"; }
run base
for cfg in "DCF_WGRAD_SIDE=96" "DCF_WGRAD_SIDE=64" "DCF_WGRAD_SIDE=128" "DCF_WGRAD_SIDE=96 DCF_WGRAD_SIDE_MODE=stride" "DCF_WGRAD_SIDE=128 DCF_WGRAD_SIDE_MODE=stride" "DCF_WGRAD_SIDE=96 DCF_WGRAD_SIDE_N=4" "DCF_WGRAD_SIDE=256"; do ( export $cfg; run "$cfg" ); done
run base
