#!/bin/bash
# copy one profile round (tools/_prof_all.sh <tag>) from gpurun_out/ into profiles/ under the names the earlier rounds use
T=$1
cd "$(dirname "$0")/.."
for suf in "" _cfg4 _cfg5_fp8; do
  d=gpurun_out/prof_$T$suf; p=${suf#_}; [ -n "$p" ] && p=${p}_
  [ -d $d ] || continue
  cp $d/bench_line.json profiles/${T}_${p}bench_line.json
  cp $d/bench_line_under_rocprof.json profiles/${T}_${p}bench_line_under_rocprof.json
  cp $d/kernel_stats.csv profiles/${T}_${p}$( [ -z "$p" ] && echo bench_ )kernel_stats.csv
  cp $d/pmc_traffic.csv profiles/${T}_${p}pmc_traffic.csv
  cp $d/pmc_traffic.meta.json profiles/${T}_${p}pmc_traffic.meta.json
done
for f in gpurun_out/${T}_*_bench_line.json; do [ -f "$f" ] && cp $f profiles/; done
[ -f gpurun_out/${T}_pytest_gpu.log ] && cp gpurun_out/${T}_pytest_gpu.log profiles/
ls profiles | grep "^$T" | tr '\n' ' '
