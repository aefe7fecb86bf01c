#!/bin/bash
# full GPU check of the current tree: tests, smoke, bench
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02e_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02e_smoke.log 2>&1
python tools/attn_bench.py > gpurun_out/r02e_attn.log 2>&1
python bench.py --steps 5 --warmup 2 > gpurun_out/r02e_bench.log 2>&1
tail -4 gpurun_out/r02e_pytest.log; tail -2 gpurun_out/r02e_smoke.log; tail -5 gpurun_out/r02e_attn.log; tail -c 3000 gpurun_out/r02e_bench.log
