cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q 2>&1 | tail -4
echo "=== per-layer times, wg1"; python3 tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids | head -9
echo "=== per-layer times, generic (DCF_WGRAD1S=0)"; DCF_WGRAD1S=0 python3 tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids | head -9
for v in new old new old; do
  if [ $v = old ]; then export DCF_WGRAD1S=0; else unset DCF_WGRAD1S; fi
  python3 bench.py --no-cpu-baseline --no-from-host 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k['kernel']:k['ms_per_step'] for k in d['kernel_breakdown']}
print('$v', d['value'], d['ms_per_step'], {k:v for k,v in kb.items() if 'wgrad' in k})"
done
