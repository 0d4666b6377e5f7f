cd "$GRAFT_REPO_ROOT"
CFG4="--dtype f16 --batch 4 --points 120000 --knn 5 --image-stream resnet50"
run() {
  python3 bench.py --no-cpu-baseline --no-from-host --no-roofline $2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
}
run cfg4_base "$CFG4"
DCF_IGEMM_DMA=0 run cfg4_dma0 "$CFG4"
DCF_IGEMM_DMA=2 run cfg4_dma2 "$CFG4"
DCF_IGEMM_DMA=3 run cfg4_dma3 "$CFG4"
run cfg4_base "$CFG4"
run cfg2_base ""
DCF_IGEMM_DMA=2 run cfg2_dma2 ""
DCF_IGEMM_DMA=3 run cfg2_dma3 ""
run cfg2_base ""
