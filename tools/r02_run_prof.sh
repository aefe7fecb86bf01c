#!/bin/bash
# profiles of the final build (tag $1, default r02o): PMC passes (attention variants 1 / 3 / 4, GEMM variant 8 at the bench's exemplar
# batch), rocprofv3 kernel stats of the bench command
T=${1:-r02o}
VARIANTS="1 3 4" bash tools/pmc_attn.sh gpurun_out/${T}_pmc_attn > gpurun_out/${T}_pmc_attn.log 2>&1
bash tools/pmc_gemm.sh 8 gpurun_out/${T}_pmc_gemm 768 > gpurun_out/${T}_pmc_gemm.log 2>&1
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample-classes 0 > $R/gpurun_out/${T}_stats_bench.log 2>&1
cd $R
python3 - <<PY
import json
d=json.load(open("gpurun_out/${T}_pmc_attn/summary.json"))
for k,v in d.items(): print(k, {n: v[n] for n in ("duration_us_under_pmc","SQ_ACTIVE_INST_VALU","SQ_VALU_MFMA_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_INSTS_VALU","FETCH_SIZE","WRITE_SIZE","hbm_bytes_per_launch") if n in v})
d=json.load(open("gpurun_out/${T}_pmc_gemm/summary.json"))
for k,v in d["kernels"].items():
    if "<7, 8" in k: print(k, v)
PY
f=$(find gpurun_out/${T}_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${T}_kernel_stats.csv; head -14 $f | cut -c1-220; tail -1 gpurun_out/${T}_stats_bench.log | cut -c1-1800
