#!/bin/bash
for a in 0 16; do echo "abl $a"; OVMR_FQ_ABL=$a timeout 200 python tools/fused_qkv_bench.py --batches 64 775 --reps 10 2>&1 | grep images | cut -c1-230; done
