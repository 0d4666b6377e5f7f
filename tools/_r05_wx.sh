cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "wgrad" 2>&1 | tail -2
echo "=== per-layer times, new mapping"; python3 tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids | head -8
echo "=== per-layer times, old mapping (DCF_WGRAD_XCD_MIN9=48)"; DCF_WGRAD_XCD_MIN9=48 python3 tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids | head -8
echo "=== step A/B: base = new mapping, opt = old"
bash tools/_ab.sh DCF_WGRAD_XCD_MIN9=48
TAG=r05p bash tools/_r05_wt.sh
