python -m pytest tests/test_gpu_elementwise.py -m gpu -x -q -k "relu_mask_rowscale or rowscale" 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-batch-sweep --input resident --no-other-leg --steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
for k in d['kernel_breakdown']:
    if k['kernel'] in ('relu_mask_rowscale_bwd','relu_bwd_chansum','resize_bwd','point_sample_bwd'): print(k)"
