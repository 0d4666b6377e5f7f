# bash tools/_sweep.sh OPT=VAL ... : cfg2 bench line with each option set in turn (env), a base run before and after
run() { python bench.py --no-cpu-baseline --no-from-host --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % '$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"; }
run base
for o in "$@"; do ( export $o; run $o ); done
run base
