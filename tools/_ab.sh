# same-box A/B of an option: bash tools/_ab.sh DCF_KNN_MERGED_SEARCH=0 [bench args]
opt=$1; shift
for r in 1 2; do
  for v in base opt; do
    if [ $v = opt ]; then export $opt; else unset ${opt%%=*}; fi
    python bench.py --no-cpu-baseline --no-from-host --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done
