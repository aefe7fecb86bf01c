#!/bin/bash
# attention A/B + PMC, GEMM v8 PMC, rocprofv3 stats of the bench
python -m pytest tests/test_hip_kernels.py -q -x -k "attention" 2>&1 | tail -3 > gpurun_out/r02f_attn_tests.log
python tools/attn_bench.py > gpurun_out/r02f_attn.log 2>&1
bash tools/pmc_attn.sh gpurun_out/r02f_pmc_attn > gpurun_out/r02f_pmc_attn.log 2>&1
bash tools/pmc_gemm.sh 8 gpurun_out/r02f_pmc_gemm 512 > gpurun_out/r02f_pmc_gemm.log 2>&1
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f_stats -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample-classes 0 > $R/gpurun_out/r02f_stats_bench.log 2>&1
cd $R
tail -3 gpurun_out/r02f_attn_tests.log; cat gpurun_out/r02f_attn.log | tail -4
python3 - <<PY
import json
d=json.load(open("gpurun_out/r02f_pmc_attn/summary.json"))
for k,v in d.items(): print(k, {n: v[n] for n in ("duration_us_under_pmc","SQ_BUSY_CYCLES","SQ_ACTIVE_INST_VALU","SQ_VALU_MFMA_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_INSTS_VALU") if n in v})
d=json.load(open("gpurun_out/r02f_pmc_gemm/summary.json"))
for k,v in d["kernels"].items(): print(k, {n: v.get(n) for n in ("duration_us_under_pmc","hbm_bytes_per_launch","FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum","TCC_MISS_sum","SQ_VALU_MFMA_BUSY_CYCLES","SQ_BUSY_CYCLES","GRBM_GUI_ACTIVE")})
PY
find gpurun_out/r02f_stats -name "*kernel_stats.csv" | head -2; f=$(find gpurun_out/r02f_stats -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200; tail -1 gpurun_out/r02f_stats_bench.log | cut -c1-300
