#!/bin/bash
# round-2 GEMM experiments (one gpurun call): correctness of variant 8, then same-process A/B timings
set -x
python -m pytest tests/test_hip_kernels.py -q -x -k "gemm" 2>&1 | tail -5 > gpurun_out/r02c_kernels.log
python tools/gemm_bench.py --batch 512 --variants 6 8 > gpurun_out/r02c_gemm_v6_v8.log 2>&1
OVMR_A_NT=3 python tools/gemm_bench.py --batch 512 --variants 6 8 > gpurun_out/r02c_gemm_ant.log 2>&1
OVMR_N_GROUP=4 python tools/gemm_bench.py --batch 512 --variants 6 8 > gpurun_out/r02c_gemm_g4.log 2>&1
OVMR_N_GROUP=4 OVMR_A_NT=3 python tools/gemm_bench.py --batch 512 --variants 6 8 > gpurun_out/r02c_gemm_g4_ant.log 2>&1
OVMR_N_GROUP=6 OVMR_A_NT=3 python tools/gemm_bench.py --batch 512 --variants 6 8 > gpurun_out/r02c_gemm_g6_ant.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm 8 > gpurun_out/r02c_bench_v8.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm 6 --batch 544 --classes-per-batch 238 > gpurun_out/r02c_bench_v6_b544.log 2>&1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm 8 --batch 544 --classes-per-batch 238 > gpurun_out/r02c_bench_v8_b544.log 2>&1
tail -3 gpurun_out/r02c_kernels.log
for f in gpurun_out/r02c_gemm_*.log; do echo "== $f"; grep -E "^(qkv|out_proj|c_fc|c_proj|qkv_ln|c_fc_ln|out_proj_st|c_proj_st) " $f | cut -c1-260; done
for f in gpurun_out/r02c_bench_*.log; do echo "== $f"; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['achieved'], d['phases'])"; done
python tools/cpu_baseline_probe.py --procs 1 4 16 > gpurun_out/r02c_cpu_probe.log 2>&1
cut -c1-400 gpurun_out/r02c_cpu_probe.log
