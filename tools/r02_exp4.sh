#!/bin/bash
# full GPU suite + bench with attention variant 3 and 4 + attention ablation table (experiment build)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02n_pytest.log 2>&1; tail -2 gpurun_out/r02n_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02n_bench_attn3.log 2>&1; grep '^{"metric' gpurun_out/r02n_bench_attn3.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --attn 4 > gpurun_out/r02n_bench_attn4.log 2>&1; grep '^{"metric' gpurun_out/r02n_bench_attn4.log | cut -c1-200
timeout 300 python tools/attn_bench.py --only image --variants 1 3 4 401 402 404 412 420 436 460 464 468 524 2>&1 | tee gpurun_out/r02n_attn_ablations.log | tail -1
