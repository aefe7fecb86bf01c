#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02j_pytest.log
python tools/xval_bench.py --classes 10000 --shots 8 > gpurun_out/r02j_xval_c10000_s8.log 2>&1
python tools/xval_bench.py --classes 1000 --shots 16 > gpurun_out/r02j_xval_c1000_s16.log 2>&1
OVMR_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --classes 200 --queries 1024 --no-cpu-baseline > gpurun_out/r02j_bench_gloo2.log 2>&1
python bench.py --steps 2 --warmup 1 --classes 200 --queries 1024 --no-cpu-baseline > gpurun_out/r02j_bench_n1_small.log 2>&1
tail -3 gpurun_out/r02j_pytest.log; tail -1 gpurun_out/r02j_xval_c10000_s8.log; tail -1 gpurun_out/r02j_xval_c1000_s16.log
for f in gpurun_out/r02j_bench_gloo2.log gpurun_out/r02j_bench_n1_small.log; do echo == $f; grep "^{\"metric" $f | cut -c1-330 || tail -5 $f; done; tail -5 gpurun_out/r02j_bench_gloo2.log | cut -c1-300
