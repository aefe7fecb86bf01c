#!/bin/bash
# round 5, pass d: the default bench line with the other configurations behind it, c4 alone with kernel stats
T=r05d; R=$(pwd); mkdir -p gpurun_out/${T}_keep
timeout 1700 python bench.py > gpurun_out/${T}_bench.log 2>&1; echo "bench rc $?"; grep '^{"metric' gpurun_out/${T}_bench.log > gpurun_out/${T}_keep/${T}_bench_n1.json; tail -c 3000 gpurun_out/${T}_bench.log
cd /tmp; export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats_c4 -- python3 $R/bench.py --preset c4 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${T}_stats_c4.log 2>&1
cd $R
f=$(find gpurun_out/${T}_stats_c4 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${T}_keep/${T}_rocprofv3_kernel_stats_bench_c4.csv; head -8 $f | cut -c1-200
grep '^{"metric' gpurun_out/${T}_stats_c4.log > gpurun_out/${T}_keep/${T}_bench_c4_under_rocprof.json
rm -rf gpurun_out/${T}_stats_c4
