#!/bin/bash
# One GPU pass of a build (tag $1, e.g. r03e): the -m gpu suite, smoke, the default bench line, the same bench under
# `rocprofv3 --kernel-trace --stats`, and the PMC passes of the dominant GEMM (product library, the variant + QuickGELU form the
# engine launches: 108) and of attention.  Results land in gpurun_out/<tag>_*; the summaries worth keeping are copied to profiles/.
# usage: tools/run_prof.sh <tag> [skip-tests]
T=${1:-r03x}; R=$(pwd)
mkdir -p gpurun_out profiles
if [ -z "$2" ]; then
  timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; tail -2 gpurun_out/${T}_pytest.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
fi
bash tools/pmc_gemm.sh 108 gpurun_out/${T}_pmc_gemm 775 > gpurun_out/${T}_pmc_gemm.log 2>&1
cp gpurun_out/${T}_pmc_gemm/summary.json profiles/${T}_pmc_gemm_v108.json
VARIANTS="1 3" bash tools/pmc_attn.sh gpurun_out/${T}_pmc_attn > gpurun_out/${T}_pmc_attn.log 2>&1
cp gpurun_out/${T}_pmc_attn/summary.json profiles/${T}_pmc_attn.json
timeout 900 python bench.py > gpurun_out/${T}_bench.log 2>&1; grep '^{"metric' gpurun_out/${T}_bench.log > profiles/${T}_bench_n1.json; cut -c1-200 profiles/${T}_bench_n1.json
# round 4: the small-shard configuration, the 8-rank projection and the one-rank RCCL line of the same build on the same box
timeout 600 python bench.py --preset c2 --no-cpu-baseline > gpurun_out/${T}_c2.log 2>&1; grep '^{"metric' gpurun_out/${T}_c2.log > profiles/${T}_bench_c2.json
timeout 900 python bench.py --emulate-world 8 --no-cpu-baseline > gpurun_out/${T}_emu8.log 2>&1; grep '^{"metric' gpurun_out/${T}_emu8.log > profiles/${T}_emulated_world8.json
timeout 900 python bench.py --force-dist --no-cpu-baseline > gpurun_out/${T}_fd.log 2>&1; grep '^{"metric' gpurun_out/${T}_fd.log > profiles/${T}_bench_n1_force_dist_rccl.json
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample-classes 0 > $R/gpurun_out/${T}_stats_bench.log 2>&1
cd $R
f=$(find gpurun_out/${T}_stats -name "*kernel_stats.csv" | head -1); cp $f profiles/${T}_rocprofv3_kernel_stats_bench_full.csv; head -6 $f | cut -c1-200
grep '^{"metric' gpurun_out/${T}_stats_bench.log > profiles/${T}_bench_under_rocprof.json
python3 tools/dominant_by_grid.py $(find gpurun_out/${T}_stats -name "*kernel_trace.csv" | head -1) profiles/${T}_bench_under_rocprof.json profiles/${T}_dominant_kernel_by_grid.json > /dev/null
rm -rf gpurun_out/${T}_pmc_gemm/*/ gpurun_out/${T}_pmc_attn/*/ gpurun_out/${T}_stats     # raw counter / trace CSVs: large, the summaries are kept
mkdir -p gpurun_out/${T}_keep; cp profiles/${T}_* gpurun_out/${T}_keep/     # gpurun merges only gpurun_out/ back: copy gpurun_out/<tag>_keep/* into profiles/ afterwards
python3 - <<PY
import json
d = json.load(open("profiles/${T}_bench_n1.json"))
r = d["roofline"]
for f, k in (("bench_c2", "value"), ("emulated_world8", "projected_speedup"), ("bench_n1_force_dist_rccl", "value")):
    try: print(f, json.load(open("profiles/${T}_%s.json" % f))[k])
    except Exception as e: print(f, "failed", e)
print("bench:", d["value"], "img/s; c_fc", r["avg_launch_us"], "us, frac", r["frac"], "mfma_busy", r.get("mfma_busy_frac"), "lds_conflict", r.get("lds_conflict_frac"), "hbm GB/s", r.get("hbm_gbps"))
d = json.load(open("profiles/${T}_bench_under_rocprof.json"))
print("under rocprof:", d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_per_step"], d["roofline"]["frac"])
PY
