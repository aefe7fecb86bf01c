#!/bin/bash
# round-4 evidence pass: projections for 2 / 4 / 8 ranks, the one-rank RCCL line, preset lines, per-shape PMC of the residual projections
T=${1:-r04h}; R=$(pwd); mkdir -p gpurun_out/${T}_keep
timeout 600 python -m pytest tests -m gpu -q -k "last_block_projects or encoder_chunk or split_k or text_groups or proves_its" > gpurun_out/${T}_pytest_sel.log 2>&1; tail -3 gpurun_out/${T}_pytest_sel.log
for N in 2 4 8; do
  timeout 900 python bench.py --emulate-world $N --no-cpu-baseline > gpurun_out/${T}_emu$N.log 2>&1; grep '^{"metric' gpurun_out/${T}_emu$N.log > gpurun_out/${T}_keep/${T}_emulated_world$N.json
done
timeout 900 python bench.py --force-dist --no-cpu-baseline > gpurun_out/${T}_fd.log 2>&1; grep '^{"metric' gpurun_out/${T}_fd.log > gpurun_out/${T}_keep/${T}_bench_n1_force_dist_rccl.json
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${T}_b.log 2>&1; grep '^{"metric' gpurun_out/${T}_b.log > gpurun_out/${T}_keep/${T}_bench_n1_nocpu.json
timeout 900 python bench.py --preset c2 --no-cpu-baseline > gpurun_out/${T}_c2.log 2>&1; grep '^{"metric' gpurun_out/${T}_c2.log > gpurun_out/${T}_keep/${T}_bench_c2.json
timeout 900 python bench.py --preset c3 --no-cpu-baseline --overlap 1 > gpurun_out/${T}_c3.log 2>&1; grep '^{"metric' gpurun_out/${T}_c3.log > gpurun_out/${T}_keep/${T}_bench_c3_ov1.json
timeout 900 python bench.py --preset c3 --no-cpu-baseline --overlap 0 > gpurun_out/${T}_c3o.log 2>&1; grep '^{"metric' gpurun_out/${T}_c3o.log > gpurun_out/${T}_keep/${T}_bench_c3_ov0.json
timeout 900 python bench.py --preset c5 --no-cpu-baseline > gpurun_out/${T}_c5.log 2>&1; grep '^{"metric' gpurun_out/${T}_c5.log > gpurun_out/${T}_keep/${T}_bench_c5.json
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${T}_keep/*.json")):
    try:
        d = json.load(open(f))
        if d.get("projection"): print(f.split("/")[-1], d["whole_job_ms_one_rank"], d["slowest_rank_ms"], d["projected_speedup"])
        else: print(f.split("/")[-1], d["value"], d["ms_per_step"], d["phases"]["generation_images_per_s_rank0"], d["phases"]["inference_images_per_s_rank0"], (d.get("dist") or {}).get("rccl_version"), (d.get("dist") or {}).get("devices_seen"))
    except Exception as e: print(f, "failed", e)
PY
bash tools/pmc_gemm.sh 108 gpurun_out/${T}_pmc_gemm 775 > gpurun_out/${T}_pmc_gemm.log 2>&1
cp gpurun_out/${T}_pmc_gemm/summary.json gpurun_out/${T}_keep/${T}_pmc_gemm_v108.json
python - <<PY
import json
d = json.load(open("gpurun_out/${T}_keep/${T}_pmc_gemm_v108.json"))
for k, c in d["kernels"].items():
    if "shape=" in k or "<7, 8" in k:
        print(k, {x: c.get(x) for x in ("duration_us_under_pmc", "hbm_bytes_per_launch", "hbm_gbps", "mfma_busy_frac", "clock_ghz", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "lds_conflict_frac")})
PY
rm -rf gpurun_out/${T}_pmc_gemm/*/  # the raw counter CSVs are large; the summary is kept
