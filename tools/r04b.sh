#!/bin/bash
# round-4 pass: [tests,] c2 line, 8-rank projection, default line (no CPU leg)
T=${1:-r04b}; R=$(pwd); mkdir -p gpurun_out
if [ -z "$2" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; tail -15 gpurun_out/${T}_pytest.log
fi
timeout 600 python bench.py --preset c2 --no-cpu-baseline > gpurun_out/${T}_c2.log 2>&1; grep '^{"metric' gpurun_out/${T}_c2.log > gpurun_out/${T}_bench_c2.json; tail -2 gpurun_out/${T}_c2.log | cut -c1-300
timeout 900 python bench.py --emulate-world 8 --no-cpu-baseline > gpurun_out/${T}_emu8.log 2>&1; grep '^{"metric' gpurun_out/${T}_emu8.log > gpurun_out/${T}_emulated_world8.json; tail -2 gpurun_out/${T}_emu8.log | cut -c1-400
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.log 2>&1; grep '^{"metric' gpurun_out/${T}_bench.log > gpurun_out/${T}_bench_n1.json; tail -2 gpurun_out/${T}_bench.log | cut -c1-300
python - <<PY
import json
for f in ("bench_c2", "bench_n1"):
    try:
        d = json.load(open("gpurun_out/${T}_%s.json" % f)); print(f, d["value"], d["ms_per_step"], d["phases"]["generation_images_per_s_rank0"], d["phases"]["inference_images_per_s_rank0"])
    except Exception as e: print(f, "failed", e)
try:
    d = json.load(open("gpurun_out/${T}_emulated_world8.json")); print("emu8", d["whole_job_ms_one_rank"], d["slowest_rank_ms"], d["projected_speedup"], [p["generation_ms"] for p in d["per_rank"]][:3])
except Exception as e: print("emu failed", e)
PY
