cd "$GRAFT_REPO_ROOT"
for v in new old new old; do
  if [ $v = old ]; then export DCF_WGRAD_XCD_MIN9=48; else unset DCF_WGRAD_XCD_MIN9; fi
  python3 bench.py --no-cpu-baseline --no-from-host 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k['kernel']:k['ms_per_step'] for k in d['kernel_breakdown']}
print('$v', d['value'], d['ms_per_step'], {k:v for k,v in kb.items() if 'wgrad' in k})"
done
