#!/bin/bash
# exemplar encoder batch against tile-round quantisation (256 CUs): bench.py at several --batch values
for b in "$@"; do
  timeout 400 python bench.py --no-cpu-baseline --batch $b 2>/dev/null | grep '^{"metric' > /tmp/b_$b.json
  python - "$b" <<'PY'
import json, sys
d = json.load(open(f"/tmp/b_{sys.argv[1]}.json"))
print(sys.argv[1], d["value"], d["phases"]["generation_images_per_s_rank0"], d["phases"]["inference_images_per_s_rank0"], d["roofline"]["frac"], flush=True)
PY
done
