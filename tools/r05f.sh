#!/bin/bash
T=r05f; mkdir -p gpurun_out/${T}_keep
timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 -s > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed|headline job:|c4 at ViT" gpurun_out/${T}_pytest.log | tail -8; grep -A14 "slowest" gpurun_out/${T}_pytest.log | cut -c1-150
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
