cd "$GRAFT_REPO_ROOT"
# A/B of the kind-1 launches of <= 8 position tiles with the pixel tile two stages ahead (RS_DX2, default) against one (RS_DX2=0)
timeout 900 python3 -m pytest tests/test_gpu_conv.py -m gpu -x -q -k "conv_fwd or conv_dgrad or sixteen" 2>&1 | tail -2
echo "=== DX=2"; python3 tools/conv_bench.py 2>&1 | grep -E "^l2 |^l4 |^conv3|^i1 |^i2 "
echo "=== DX=1 (DCF_RS_DX2=0)"; DCF_RS_DX2=0 python3 tools/conv_bench.py 2>&1 | grep -E "^l2 |^l4 |^conv3|^i1 |^i2 "
bash tools/_ab.sh DCF_RS_DX2=0
