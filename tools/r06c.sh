#!/bin/bash
# round 6, second GPU pass: the whole -m gpu suite with durations, smoke, then the sharded-job evidence again (sync removed, pack / unpack kernels)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r06c_pytest.log 2>&1; tail -32 gpurun_out/r06c_pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06c_smoke.log 2>&1; tail -1 gpurun_out/r06c_smoke.log
timeout 900 python bench.py --emulate-world 8 --no-cpu-baseline --presets 0 > gpurun_out/r06c_emu8.log 2>&1; grep '^{"metric' gpurun_out/r06c_emu8.log > gpurun_out/r06c_emulated_world8.json
timeout 600 python bench.py --preset c2 --no-cpu-baseline --presets 0 --steps 5 --warmup 2 > gpurun_out/r06c_c2.log 2>&1; grep '^{"metric' gpurun_out/r06c_c2.log > gpurun_out/r06c_bench_c2.json
timeout 600 python bench.py --preset c1 --steps 3 --warmup 1 > gpurun_out/r06c_c1.log 2>&1; grep '^{"metric' gpurun_out/r06c_c1.log > gpurun_out/r06c_bench_c1.json
timeout 600 python tools/sync_audit.py > gpurun_out/r06c_sync_audit.log 2>&1; tail -4 gpurun_out/r06c_sync_audit.log
timeout 600 python tools/head_bench.py --exp > gpurun_out/r06c_head_relaxed.log 2>&1
timeout 600 python tools/head_bench.py --acqrel > gpurun_out/r06c_head_acqrel.log 2>&1
paste -d'\n' gpurun_out/r06c_head_relaxed.log gpurun_out/r06c_head_acqrel.log | grep one_launch | cut -c1-160
python3 - <<PY
import json
for f, ks in (("emulated_world8", ("projected_speedup", "whole_job_ms_one_rank", "whole_job_ms_timed_before_the_shards", "whole_job_ms_timed_after_the_shards", "slowest_rank_ms")),
              ("bench_c2", ("value",)), ("bench_c1", ("value", "gpu_over_cpu"))):
    try:
        d = json.load(open("gpurun_out/r06c_%s.json" % f))
        print(f, {k: d.get(k) for k in ks}, [p["ms_per_step"] for p in d.get("per_rank", [])], (d.get("phases") or {}).get("generation_images_per_s_rank0"), (d.get("cpu_baseline") or {}).get("sample", "")[:300])
    except Exception as e:
        print(f, "failed", e)
PY
