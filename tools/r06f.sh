#!/bin/bash
for a in 0 1 2 4 8 6 7 15; do echo "abl $a"; OVMR_FQ_ABL=$a timeout 200 python tools/fused_qkv_bench.py --batches 775 --reps 5 2>&1 | grep images | cut -c1-220; done
