#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "async_file or cli_generate or trainer_shim or generate_classifier_vs_golden" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_hip_distributed.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python bench.py --emulate-world 8 --steps 5 --warmup 2 --no-cpu-baseline --presets 0 > gpurun_out/r06g_emu8.log 2>&1; grep '^{"metric' gpurun_out/r06g_emu8.log > gpurun_out/r06g_emulated_world8.json
timeout 600 python bench.py --preset c2 --no-cpu-baseline --presets 0 --steps 5 --warmup 2 > gpurun_out/r06g_c2.log 2>&1; grep '^{"metric' gpurun_out/r06g_c2.log > gpurun_out/r06g_bench_c2.json
timeout 300 python tools/query_launch_probe.py > gpurun_out/r06g_query_launch_probe.log 2>&1; tail -2 gpurun_out/r06g_query_launch_probe.log
python3 - <<PY
import json
d = json.load(open("gpurun_out/r06g_emulated_world8.json"))
print({k: d[k] for k in ("projected_speedup", "whole_job_ms_one_rank", "slowest_rank_ms")}, [p["ms_per_step"] for p in d["per_rank"]])
d = json.load(open("gpurun_out/r06g_bench_c2.json")); print("c2", d["value"], d["phases"]["generation_images_per_s_rank0"], d["phases"]["files_join_ms_after_generation_alone"])
PY
