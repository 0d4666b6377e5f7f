# kernel_breakdown rows matching a pattern: bash tools/_kb.sh PATTERN [bench args]
pat=$1; shift
python bench.py --no-cpu-baseline --no-from-host --steps 30 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
for k in d.get('kernel_breakdown', []):
    if '$pat' in k['kernel']: print(k)"
