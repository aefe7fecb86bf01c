#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -x -q > gpurun_out/r02q_kernels.log 2>&1; tail -3 gpurun_out/r02q_kernels.log
OVMR_NO_SPLIT=1 timeout 600 python tools/gemm_bench.py --batch 256 --variants 8 2>&1 | grep -v amdgpu > gpurun_out/r02q_gemm_b256_nosplit.log
timeout 600 python tools/gemm_bench.py --batch 256 --variants 8 2>&1 | grep -v amdgpu > gpurun_out/r02q_gemm_b256_split.log
paste -d'\n' gpurun_out/r02q_gemm_b256_nosplit.log gpurun_out/r02q_gemm_b256_split.log | cut -c1-150 | head -24
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02q_bench.log 2>&1; grep '^{"metric' gpurun_out/r02q_bench.log | cut -c1-230
