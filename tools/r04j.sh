#!/bin/bash
# round-4: new parity tests, multi-rank lines with ranks sharing the one GPU (gloo staging), experiment build check
T=${1:-r04j}; R=$(pwd); mkdir -p gpurun_out/${T}_keep
timeout 3000 python -m pytest tests -m gpu -q -x -k "c2_vit_b16 or c2_hundred or c5_vit_l14 or cli_generate or headline or distributed or proves_its or multi_process or rccl" > gpurun_out/${T}_pytest_sel.log 2>&1; tail -5 gpurun_out/${T}_pytest_sel.log
for N in 2 4; do
  OVMR_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --no-cpu-baseline > gpurun_out/${T}_gloo$N.log 2>&1
  grep '^{"metric' gpurun_out/${T}_gloo$N.log > gpurun_out/${T}_keep/${T}_bench_gloo_${N}ranks_one_gpu.json; tail -2 gpurun_out/${T}_gloo$N.log | cut -c1-300
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${T}_keep/*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("dist"))
    except Exception as e: print(f, "failed", e)
PY
