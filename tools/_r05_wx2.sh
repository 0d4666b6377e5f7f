cd "$GRAFT_REPO_ROOT"
echo "=== per-layer times, new mapping"; python3 tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids | head -4
echo "=== per-layer times, old mapping (DCF_WGRAD_XCD_MIN9=48)"; DCF_WGRAD_XCD_MIN9=48 python3 tools/wgrad_layers.py 2>&1 | grep -v amdgpu.ids | head -4
