#!/bin/bash
# One GPU pass of a build (tag $1, e.g. r03e): the -m gpu suite, smoke, the default bench line, the same bench under
# `rocprofv3 --kernel-trace --stats`, and the PMC passes of the dominant GEMM (product library, the variant + QuickGELU form the
# engine launches: 108) and of attention.  Results land in gpurun_out/<tag>_*; the summaries worth keeping are copied to profiles/.
# usage: tools/run_prof.sh <tag> [skip-tests]
T=${1:-r03x}; R=$(pwd)
mkdir -p gpurun_out profiles
if [ -z "$2" ]; then
  timeout 3000 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/${T}_pytest.log 2>&1; tail -14 gpurun_out/${T}_pytest.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
fi
bash tools/pmc_gemm.sh 108 gpurun_out/${T}_pmc_gemm 775 > gpurun_out/${T}_pmc_gemm.log 2>&1
cp gpurun_out/${T}_pmc_gemm/summary.json profiles/${T}_pmc_gemm_v108.json
VARIANTS="1 3" bash tools/pmc_attn.sh gpurun_out/${T}_pmc_attn > gpurun_out/${T}_pmc_attn.log 2>&1
cp gpurun_out/${T}_pmc_attn/summary.json profiles/${T}_pmc_attn.json
timeout 900 python bench.py > gpurun_out/${T}_bench.log 2>&1; grep '^{"metric' gpurun_out/${T}_bench.log > profiles/${T}_bench_n1.json; cut -c1-200 profiles/${T}_bench_n1.json
# (the default bench line above carries c2 / c3 / c4 / c5 as `presets`)  The 8-rank projection, the one-rank RCCL line and the 2-rank line
# over gloo on this one GPU (launched exactly as the driver launches N > 1: per-rank times and the collectives in `dist`) of the same build, same box
timeout 900 python bench.py --emulate-world 8 --steps 5 --warmup 2 --no-cpu-baseline --presets 0 > gpurun_out/${T}_emu8.log 2>&1; grep '^{"metric' gpurun_out/${T}_emu8.log > profiles/${T}_emulated_world8.json
timeout 900 python bench.py --force-dist --no-cpu-baseline --presets 0 > gpurun_out/${T}_fd.log 2>&1; grep '^{"metric' gpurun_out/${T}_fd.log > profiles/${T}_bench_n1_force_dist_rccl.json
for n in 2 4; do   # the driver's own N > 1 command, ranks sharing this one GPU over gloo (OVMR_DIST_BACKEND): per-rank times and the collectives in `dist`
  OVMR_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 > gpurun_out/${T}_gloo$n.log 2>&1; grep '^{"metric' gpurun_out/${T}_gloo$n.log > profiles/${T}_bench_gloo_${n}ranks_one_gpu.json
done
timeout 300 python tools/sync_audit.py > profiles/${T}_sync_audit.log 2>&1
timeout 600 python tools/head_bench.py > profiles/${T}_head_bench.log 2>&1
timeout 600 python tools/attn_bench.py --variants 1 3 > profiles/${T}_attn_bench.log 2>&1
bash tools/pmc_attn_l577.sh gpurun_out/${T}_pmc_attn_l577 > gpurun_out/${T}_pmc_attn_l577.log 2>&1; cp gpurun_out/${T}_pmc_attn_l577/summary.json profiles/${T}_pmc_attn_l577.json
bash tools/pmc_head.sh gpurun_out/${T}_pmc_head > profiles/${T}_pmc_head.log 2>&1; cp gpurun_out/${T}_pmc_head/summary.json profiles/${T}_pmc_head.json; rm -rf gpurun_out/${T}_pmc_head/*/
timeout 900 python tools/race_screen.py --reps 200 > profiles/${T}_race_screen.log 2>&1
timeout 600 python tools/resize_sweep.py > profiles/${T}_resize_sweep.log 2>&1
timeout 900 python tools/pipeline_bench.py --workers 16 --host-resize > gpurun_out/${T}_pipeline.log 2>&1; grep -E "^input pipeline|usable" gpurun_out/${T}_pipeline.log > profiles/${T}_pipeline_bench.log
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample-classes 0 --presets 0 > $R/gpurun_out/${T}_stats_bench.log 2>&1
cd $R
f=$(find gpurun_out/${T}_stats -name "*kernel_stats.csv" | head -1); cp $f profiles/${T}_rocprofv3_kernel_stats_bench_full.csv; head -6 $f | cut -c1-200
grep '^{"metric' gpurun_out/${T}_stats_bench.log > profiles/${T}_bench_under_rocprof.json
python3 tools/dominant_by_grid.py $(find gpurun_out/${T}_stats -name "*kernel_trace.csv" | head -1) profiles/${T}_bench_under_rocprof.json profiles/${T}_dominant_kernel_by_grid.json > /dev/null
rm -rf gpurun_out/${T}_pmc_gemm/*/ gpurun_out/${T}_pmc_attn/*/ gpurun_out/${T}_pmc_attn_l577/*/ gpurun_out/${T}_stats     # raw counter / trace CSVs: large, the summaries are kept
mkdir -p gpurun_out/${T}_keep; cp profiles/${T}_* gpurun_out/${T}_keep/     # gpurun merges only gpurun_out/ back: copy gpurun_out/<tag>_keep/* into profiles/ afterwards
python3 - <<PY
import json
d = json.load(open("profiles/${T}_bench_n1.json"))
r = d["roofline"]
for f, k in (("emulated_world8", "projected_speedup"), ("bench_n1_force_dist_rccl", "value"), ("bench_gloo_2ranks_one_gpu", "value")):
    try: print(f, json.load(open("profiles/${T}_%s.json" % f))[k])
    except Exception as e: print(f, "failed", e)
print("presets:", {k: v.get("value", v.get("error")) for k, v in (d.get("presets") or {}).items()})
print("bench:", d["value"], "img/s; c_fc", r["avg_launch_us"], "us, frac", r["frac"], "mfma_busy", r.get("mfma_busy_frac"), "lds_conflict", r.get("lds_conflict_frac"), "hbm GB/s", r.get("hbm_gbps"))
d = json.load(open("profiles/${T}_bench_under_rocprof.json"))
print("under rocprof:", d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["launches_per_step"], d["roofline"]["frac"])
PY
