#!/bin/bash
# round 5, pass a: configuration 4 at its own architecture -- the bench line and the new GPU test
T=r05a; mkdir -p gpurun_out/${T}_keep
timeout 1500 python bench.py --preset c4 --no-cpu-baseline > gpurun_out/${T}_c4.log 2>&1; echo "c4 rc $?"; tail -3 gpurun_out/${T}_c4.log | cut -c1-1500
grep '^{"metric' gpurun_out/${T}_c4.log > gpurun_out/${T}_keep/${T}_bench_c4.json
timeout 2400 python -m pytest tests/test_hip_configs.py -m gpu -x -q -k "c4_vitb16" -s > gpurun_out/${T}_pytest_c4.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/${T}_pytest_c4.log
