#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --no-roofline --no-other-leg "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAGX', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min_max'])"; }
for i in 1 2 3 4 5 6; do
TAGX="resident      " run --input resident
TAGX="host workers 0" run
TAGX="host workers 2" run --loader-workers 2
TAGX="host si 1e-4  " DCF_SWITCH_INTERVAL=0.0001 run
TAGX="host si 1e-4 w2" DCF_SWITCH_INTERVAL=0.0001 run --loader-workers 2
TAGX="host si 1e-4 w4" DCF_SWITCH_INTERVAL=0.0001 run --loader-workers 4
done
