#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r02d_pytest.log
python tools/gemm_bench.py --batch 512 --variants 8 28 29 > gpurun_out/r02d_gemm_ablate.log 2>&1
python tools/gemm_bench.py --batch 256 --variants 6 8 > gpurun_out/r02d_gemm_b256.log 2>&1
python bench.py --steps 5 --warmup 2 > gpurun_out/r02d_bench.log 2>&1
tail -4 gpurun_out/r02d_pytest.log
for f in gpurun_out/r02d_gemm_ablate.log gpurun_out/r02d_gemm_b256.log; do echo "== $f"; grep -E "^(qkv|out_proj|c_fc|c_fc_bias_only|c_proj|qkv_ln|c_fc_ln|cls_|patch|text_qkv|xval) " $f | cut -c1-330; done
tail -c 2600 gpurun_out/r02d_bench.log
