cd "$GRAFT_REPO_ROOT"
# A/B of the 16-wave (8 consumer + 8 loader) form of conv_wg1.hip against its 8-wave form (WGRAD1S_L16=0)
timeout 900 python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "wgrad" 2>&1 | tail -2
echo "=== 16 waves"; python3 tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids | head -9
echo "=== 8 waves"; DCF_WGRAD1S_L16=0 python3 tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids | head -9
CFG4="--dtype f16 --batch 4 --points 120000 --knn 5 --image-stream resnet50"
run() {
  python3 bench.py --no-cpu-baseline --no-from-host $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k['kernel']:k['ms_per_step'] for k in d['kernel_breakdown']}
print('$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), {k:v for k,v in kb.items() if 'wgrad1s' in k})"
}
for r in 1 2; do
run cfg2_16 ""; DCF_WGRAD1S_L16=0 run cfg2_8 ""
done
run cfg4_16 "$CFG4"; DCF_WGRAD1S_L16=0 run cfg4_8 "$CFG4"
