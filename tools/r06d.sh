#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r06d_pytest.log 2>&1; tail -34 gpurun_out/r06d_pytest.log
sed -i 's/r06b/r06d/g' tools/r06b.sh; bash tools/r06b.sh 2>&1 | tail -3
