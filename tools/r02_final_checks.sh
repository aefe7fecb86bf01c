#!/bin/bash
# final GPU pass of the round (tag $1): whole -m gpu suite, smoke, default bench, the same bench under rocprofv3 --stats
T=${1:-r02t}
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; tail -2 gpurun_out/${T}_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -2 gpurun_out/${T}_smoke.log
timeout 900 python bench.py > gpurun_out/${T}_bench.log 2>&1; grep '^{"metric' gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_n1.json; cut -c1-260 gpurun_out/${T}_bench_n1.json
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample-classes 0 > $R/gpurun_out/${T}_stats_bench.log 2>&1
cd $R
f=$(find gpurun_out/${T}_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${T}_kernel_stats.csv; head -5 $f | cut -c1-200
grep '^{"metric' gpurun_out/${T}_stats_bench.log > gpurun_out/${T}_bench_under_rocprof.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/${T}_bench_under_rocprof.json"))
print("under rocprof:", d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_per_step"], d["roofline"]["frac"])
PY
