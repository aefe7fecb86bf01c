#!/bin/bash
# round-4 pass: [tests,] same-box A/B of the small-shard work (old = --gemm 7 --stream-text 0) on c2 and on the 8-rank projection,
# then the segment report (tools/timeline.py) of one emulated rank
T=${1:-r04c}; R=$(pwd); mkdir -p gpurun_out
if [ -z "$2" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; tail -15 gpurun_out/${T}_pytest.log
fi
for tag in old new; do
  if [ $tag = old ]; then F="--gemm 7 --stream-text 0"; else F=""; fi
  timeout 600 python bench.py --preset c2 --no-cpu-baseline $F > gpurun_out/${T}_c2_$tag.log 2>&1; grep '^{"metric' gpurun_out/${T}_c2_$tag.log > gpurun_out/${T}_bench_c2_$tag.json
  timeout 900 python bench.py --emulate-world 8 --no-cpu-baseline $F > gpurun_out/${T}_emu8_$tag.log 2>&1; grep '^{"metric' gpurun_out/${T}_emu8_$tag.log > gpurun_out/${T}_emulated_world8_$tag.json
done
python - <<PY
import json
for tag in ("old", "new"):
    try:
        d = json.load(open("gpurun_out/${T}_bench_c2_%s.json" % tag)); print(tag, "c2", d["value"], d["ms_per_step"], d["phases"]["generation_images_per_s_rank0"], d["phases"]["inference_images_per_s_rank0"])
        d = json.load(open("gpurun_out/${T}_emulated_world8_%s.json" % tag)); print(tag, "emu8", d["whole_job_ms_one_rank"], d["slowest_rank_ms"], d["projected_speedup"], [p["generation_ms"] for p in d["per_rank"]][:3])
    except Exception as e: print(tag, "failed", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${T}_trace_emu -- python3 $R/bench.py --emulate-world 8 --emulate-rank 0 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${T}_trace_emu.log 2>&1
cd $R
f=$(find gpurun_out/${T}_trace_emu -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f --out gpurun_out/${T}_timeline_emu.json > gpurun_out/${T}_timeline_emu.txt 2>&1
tail -14 gpurun_out/${T}_timeline_emu.txt | cut -c1-260
rm -rf gpurun_out/${T}_trace_emu
