#!/bin/bash
T=r05e; mkdir -p gpurun_out/${T}_keep
true
timeout 1500 python tools/pipeline_bench.py --workers 8 16 --host-resize > gpurun_out/${T}_pipeline_bench.log 2>&1; grep -E "input pipeline|cpus" gpurun_out/${T}_pipeline_bench.log
df -h /dev/shm | tail -1
