#!/bin/bash
# the other BASELINE.json configurations on the current build (one MI355X, synthetic weights / data, no CPU leg)
mkdir -p gpurun_out
run() { tag=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric' > gpurun_out/${T:-r02s}_bench_$tag.json
  python - "$tag" "gpurun_out/${T:-r02s}_bench_$tag.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["value"], d["phases"].get("encoder_tflops_e2e_algorithmic"), d["roofline"]["frac"], d["config"]["workload"], flush=True)
PY
}
run c5_vitl336 --model ViT-L/14@336px --classes 64 --shots 32 --queries 512 --batch 128 --classes-per-batch 256
run c2_100x8 --classes 100 --shots 8 --queries 1024 --batch 800 --classes-per-batch 100
run vitb32 --model ViT-B/32 --batch 1024 --classes-per-batch 256
