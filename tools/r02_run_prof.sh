#!/bin/bash
# profiles of the final build: PMC passes (attention, GEMM variant 8), rocprofv3 kernel stats of the bench command
bash tools/pmc_attn.sh gpurun_out/r02i_pmc_attn > gpurun_out/r02i_pmc_attn.log 2>&1
bash tools/pmc_gemm.sh 8 gpurun_out/r02i_pmc_gemm 512 > gpurun_out/r02i_pmc_gemm.log 2>&1
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02i_stats -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample-classes 0 > $R/gpurun_out/r02i_stats_bench.log 2>&1
cd $R
python3 - <<PY
import json
d=json.load(open("gpurun_out/r02i_pmc_attn/summary.json"))
for k,v in d.items(): print(k, {n: v[n] for n in ("duration_us_under_pmc","SQ_ACTIVE_INST_VALU","SQ_VALU_MFMA_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_INSTS_VALU","FETCH_SIZE","WRITE_SIZE") if n in v})
d=json.load(open("gpurun_out/r02i_pmc_gemm/summary.json"))
for k,v in d["kernels"].items():
    if "<7, 8" in k: print(k, v)
PY
f=$(find gpurun_out/r02i_stats -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-220; tail -1 gpurun_out/r02i_stats_bench.log | cut -c1-1800
