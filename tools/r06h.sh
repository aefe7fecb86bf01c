#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "async_file or cli_generate or trainer_shim or generate_classifier_vs_golden" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_hip_distributed.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 900 python bench.py --emulate-world 8 --steps 5 --warmup 2 --no-cpu-baseline --presets 0 > gpurun_out/r06h_emu8_$i.log 2>&1; grep '^{"metric' gpurun_out/r06h_emu8_$i.log > gpurun_out/r06h_emulated_world8_$i.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/r06h_emulated_world8_$i.json"))
print({k: d[k] for k in ("projected_speedup", "whole_job_ms_one_rank", "slowest_rank_ms")}, [p["ms_per_step"] for p in d["per_rank"]])
PY
done
