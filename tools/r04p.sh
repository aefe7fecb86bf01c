#!/bin/bash
# same-box A/B of the asynchronous classifier head: headline, c2, 8-rank projection with --async-head 0 / 1
T=${1:-r04p}; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "asynchronous_head or forward_batches or emulated_world or proves_its" 2>&1 | tail -4
for ah in 0 1; do
  timeout 900 python bench.py --no-cpu-baseline --async-head $ah 2>/dev/null | grep '^{"metric' > gpurun_out/${T}_bench_ah$ah.json
  timeout 600 python bench.py --preset c2 --no-cpu-baseline --async-head $ah 2>/dev/null | grep '^{"metric' > gpurun_out/${T}_bench_c2_ah$ah.json
  timeout 900 python bench.py --emulate-world 8 --no-cpu-baseline --async-head $ah 2>/dev/null | grep '^{"metric' > gpurun_out/${T}_emulated_world8_ah$ah.json
done
python - <<PY
import json
for ah in (0, 1):
    for f in ("bench", "bench_c2"):
        d = json.load(open("gpurun_out/${T}_%s_ah%d.json" % (f, ah))); print(ah, f, d["value"], d["ms_per_step"])
    d = json.load(open("gpurun_out/${T}_emulated_world8_ah%d.json" % ah)); print(ah, "emu8", d["whole_job_ms_one_rank"], d["slowest_rank_ms"], d["projected_speedup"], all(p["classifiers_fusion_weights_outputs_bit_equal_to_whole_job"] for p in d["per_rank"]))
PY
