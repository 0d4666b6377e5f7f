for g in off on; do python bench.py --graphs $g --no-cpu-baseline --no-batch-sweep --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graphs $g', d['value'], d['ms_per_step'], d['config']['resident_frames_per_s'], d['ms_per_step_median'])"; done
