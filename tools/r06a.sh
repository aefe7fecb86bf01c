#!/bin/bash
# round 6, first GPU pass: the new tests, then the sharded-job evidence (8-rank projection, c2, one-rank RCCL, 2 / 4 ranks over gloo
# through the driver's exact command) with the file write off the critical path
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "evaluator" > gpurun_out/r06a_t_eval.log 2>&1; tail -3 gpurun_out/r06a_t_eval.log
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "async_file or cli_generate or generate_classifier_vs_golden or trainer" --durations=10 > gpurun_out/r06a_t_parity.log 2>&1; tail -15 gpurun_out/r06a_t_parity.log
timeout 900 python bench.py --emulate-world 8 --no-cpu-baseline --presets 0 > gpurun_out/r06a_emu8.log 2>&1; grep '^{"metric' gpurun_out/r06a_emu8.log > gpurun_out/r06a_emulated_world8.json
timeout 900 python bench.py --emulate-world 8 --enc-chunk 667 --no-cpu-baseline --presets 0 > gpurun_out/r06a_emu8_c667.log 2>&1; grep '^{"metric' gpurun_out/r06a_emu8_c667.log > gpurun_out/r06a_emulated_world8_chunk667.json
timeout 600 python bench.py --preset c2 --no-cpu-baseline --presets 0 --steps 5 --warmup 2 > gpurun_out/r06a_c2.log 2>&1; grep '^{"metric' gpurun_out/r06a_c2.log > gpurun_out/r06a_bench_c2.json
timeout 600 python bench.py --force-dist --no-cpu-baseline --presets 0 > gpurun_out/r06a_fd.log 2>&1; grep '^{"metric' gpurun_out/r06a_fd.log > gpurun_out/r06a_bench_n1_force_dist_rccl.json
for n in 2 4; do
  OVMR_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 3 --warmup 1 > gpurun_out/r06a_gloo$n.log 2>&1
  grep '^{"metric' gpurun_out/r06a_gloo$n.log > gpurun_out/r06a_bench_gloo_${n}ranks_one_gpu.json
done
# what the driver's N > 1 command does on a one-GPU box: the preflight's one line, exit code 2
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 1 --warmup 0 > gpurun_out/r06a_preflight.log 2>&1; echo "preflight rc=$?" >> gpurun_out/r06a_preflight.log
python3 - <<PY
import json
for f, ks in (("emulated_world8", ("projected_speedup", "whole_job_ms_one_rank", "slowest_rank_ms")), ("emulated_world8_chunk667", ("projected_speedup", "slowest_rank_ms")),
              ("bench_c2", ("value",)), ("bench_n1_force_dist_rccl", ("value",)), ("bench_gloo_2ranks_one_gpu", ("value",)), ("bench_gloo_4ranks_one_gpu", ("value",))):
    try:
        d = json.load(open("gpurun_out/r06a_%s.json" % f))
        print(f, {k: d[k] for k in ks}, [p["ms_per_step"] for p in d.get("per_rank", [])], (d.get("phases") or {}).get("generation_images_per_s_rank0"))
    except Exception as e:
        print(f, "failed", e)
PY
grep -h "preflight rc\|bench.py:" gpurun_out/r06a_preflight.log | head -3
