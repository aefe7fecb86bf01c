#!/bin/bash
# `rocprofv3 --kernel-trace --stats` of bench.py for the non-default presets (tag $1): one kernel-stats CSV per BASELINE.json configuration.
# Results in gpurun_out/<tag>_keep/ (copy into profiles/).
T=${1:-r03r}; R=$(pwd)
mkdir -p gpurun_out/${T}_keep
cd /tmp; export TMPDIR=/tmp
for p in c1 c2 c3 c4 c5; do
  extra=""; [ $p = c5 ] && extra="--steps 1 --warmup 1"; [ $p = c4 ] && extra="--steps 2 --warmup 1"
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats_$p -- python3 $R/bench.py --preset $p --no-cpu-baseline --presets 0 $extra > $R/gpurun_out/${T}_stats_$p.log 2>&1
  f=$(find $R/gpurun_out/${T}_stats_$p -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/${T}_keep/${T}_rocprofv3_kernel_stats_bench_$p.csv
  grep '^{"metric' $R/gpurun_out/${T}_stats_$p.log > $R/gpurun_out/${T}_keep/${T}_bench_${p}_under_rocprof.json
  head -5 $f | cut -c1-170
done
