#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r06e_pytest.log 2>&1; tail -34 gpurun_out/r06e_pytest.log
bash tools/r06b.sh 2>&1 | tail -2
tools/fused_qkv_probe.bin > gpurun_out/r06e_fused_qkv_probe.log 2>&1; cat gpurun_out/r06e_fused_qkv_probe.log
for b in 21 64 775; do timeout 300 python tools/attn_bench.py --only image --batch $b --variants 3 1 2>&1 | grep "^image" | sed "s/^/batch $b /"; done | tee gpurun_out/r06e_attn_small_batches.log
