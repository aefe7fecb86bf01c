#!/bin/bash
for a in ${FQ_VARIANTS:-0 1 2 4 8 6 7 15}; do echo "abl $a: $(OVMR_FQ_ABL=$a timeout 200 python tools/fused_qkv_bench.py --batches 775 --reps 5 2>&1 | grep images | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["pair_us"], d["fused_us"])')"; done
