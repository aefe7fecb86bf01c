#!/bin/bash
# the fused in_proj + attention experiment, its versions side by side (OVMR_FQ_ABL: 0 = 8 waves, lockstep K loop; 16 = ping-pong; 32 = 16 waves;
# 64 = pairs in passes of 4 heads)
for a in ${FQ_VARIANTS:-0 32 96 0 32}; do echo "abl $a"; OVMR_FQ_ABL=$a timeout 200 python tools/fused_qkv_bench.py --batches 64 775 --reps 10 2>&1 | grep images | cut -c1-230; done
