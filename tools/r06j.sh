#!/bin/bash
# two against three query batches in flight, same box, alternating: c3 (inference at batch 256) and the headline
timeout 600 python -m pytest tests/test_hip_configs.py -m gpu -x -q -k "forward_batches or forward_split" 2>&1 | tail -2
for rep in 1 2; do for n in 2 3; do
  echo "c3 in-flight $n: $(timeout 300 python bench.py --preset c3 --in-flight $n --steps 3 --warmup 1 --no-cpu-baseline --presets 0 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done
for n in 2 3; do
  echo "headline in-flight $n: $(timeout 300 python bench.py --in-flight $n --steps 3 --warmup 1 --no-cpu-baseline --presets 0 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["phases"]["inference_images_per_s_rank0"])')"
done
