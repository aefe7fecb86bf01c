#!/bin/bash
T=r05h
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "fusion_head or graph_capturable or zeroshot" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/${T}_pytest.log
timeout 600 python tools/head_bench.py --stamps > gpurun_out/${T}_head_bench.log 2>&1; cat gpurun_out/${T}_head_bench.log | tail -9
