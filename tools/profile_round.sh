#!/bin/bash
# Profile set of a round (GPU box): un-profiled bench line, rocprofv3 kernel stats, the two PMC passes, summarised into
# profiles/<tag>_*.  Usage: bash tools/profile_round.sh r02x ["extra bench args"] [name suffix]
# rocprofv3 gets the python program itself after `--` (no shell / env hop: MI355X pool rule).
set -u
TAG=$1; EXTRA=${2:-}; SUF=${3:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG$SUF
mkdir -p $OUT
python3 bench.py $EXTRA > $OUT/bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-batch-sweep --no-from-host $EXTRA > $OUT/bench_line_under_rocprof.json 2>> $OUT/bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-batch-sweep --no-from-host $EXTRA > /dev/null 2>> $OUT/bench.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-batch-sweep --no-from-host $EXTRA > /dev/null 2>> $OUT/bench.err
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.csv
cp $(ls $OUT/trace/*/*_kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
