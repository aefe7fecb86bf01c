#!/bin/bash
# round-4 baseline pass: c2 line, 8-rank projection, kernel traces of both with the segment report of tools/timeline.py
T=${1:-r04a}; R=$(pwd); mkdir -p gpurun_out
timeout 600 python bench.py --preset c2 --no-cpu-baseline > gpurun_out/${T}_c2.log 2>&1; grep '^{"metric' gpurun_out/${T}_c2.log > gpurun_out/${T}_bench_c2.json; cut -c1-160 gpurun_out/${T}_bench_c2.json
timeout 900 python bench.py --emulate-world 8 --no-cpu-baseline > gpurun_out/${T}_emu8.log 2>&1; grep '^{"metric' gpurun_out/${T}_emu8.log > gpurun_out/${T}_emulated_world8.json; tail -3 gpurun_out/${T}_emu8.log | cut -c1-600
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${T}_trace_c2 -- python3 $R/bench.py --preset c2 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${T}_trace_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${T}_trace_emu -- python3 $R/bench.py --emulate-world 8 --emulate-rank 0 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${T}_trace_emu.log 2>&1
cd $R
for k in c2 emu; do
  f=$(find gpurun_out/${T}_trace_$k -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline.py $f --out gpurun_out/${T}_timeline_$k.json > gpurun_out/${T}_timeline_$k.txt 2>&1
  head -12 gpurun_out/${T}_timeline_$k.txt
  # keep the trace itself small enough to travel back: the last 6000 kernel rows
  (head -1 $f; tail -6000 $f) > gpurun_out/${T}_trace_${k}_tail.csv
  rm -rf gpurun_out/${T}_trace_$k
done
