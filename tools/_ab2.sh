# same-box A/B of two bench.py argument sets: bash tools/_ab2.sh "--graphs off" "--graphs on"
for r in 1 2; do
  for v in "$1" "$2"; do
    python bench.py --no-cpu-baseline --no-from-host --no-roofline $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done
