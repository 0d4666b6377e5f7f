#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --batch 8 --no-roofline --no-other-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b8', d['value'], d['ms_per_step'], d['ms_per_step_median'], d.get('slowest_step_index'), d['ms_per_step_min_max'], d['ms_first_steps'][:3])"; done
python3 bench.py --no-cpu-baseline --no-roofline --no-other-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'], d['ms_per_step_median'], d.get('slowest_step_index'), d['ms_per_step_min_max'], d['ms_first_steps'][:3])"
