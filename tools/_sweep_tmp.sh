cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for b in 120 80 60 40; do
  DCF_WGRAD3_BLOCKS=$b python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k['kernel']:k['ms_per_step'] for k in d['kernel_breakdown']}
print('blocks $b', d['value'], d['ms_per_step'], 'wgrad3g', kb.get('conv_wgrad3g_grp_bf16<2,2,2,8>'), 'finalize', kb.get('wgrad_finalize'))"
done
for b in 256 128; do
  DCF_WGRAD_BLOCKS=$b python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kb={k['kernel']:k['ms_per_step'] for k in d['kernel_breakdown']}
print('generic blocks $b', d['value'], d['ms_per_step'], 'grp22', kb.get('conv_wgrad_grp_bf16<2,2>'), 'finalize', kb.get('wgrad_finalize'))"
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02d_prof -- python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline > gpurun_out/r02d_bench_line_under_rocprof.json 2>/dev/null
ls gpurun_out/r02d_prof/*/ | head
