#!/bin/bash
# kernel trace of ONE emulated rank of eight (rank 1: no file write), cut into segments by tools/timeline.py
R=$(pwd); mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r06b_trace -- python3 $R/bench.py --emulate-world 8 --emulate-rank 1 --steps 2 --warmup 1 --no-cpu-baseline --presets 0 > $R/gpurun_out/r06b_trace.log 2>&1
cd $R
f=$(find gpurun_out/r06b_trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $f --out gpurun_out/r06b_timeline.json --top 12 > gpurun_out/r06b_timeline.txt 2>&1
tail -60 gpurun_out/r06b_timeline.txt
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last shard generation: from the last agg_input_kernel back to the preceding fill_cls run ... print the head's kernels in order with gaps
# (bench.py --emulate-world times the whole job once more behind the shards: 1 warm-up + 2 steps = the last three generations)
idx = [i for i, r in enumerate(rows) if "agg_input_kernel" in r[2]][-4]
end = next(i for i in range(idx, len(rows)) if "fusion_weights_kernel" in rows[i][2])
prev = rows[idx - 1][1]
out = open("gpurun_out/r06b_head_kernels.txt", "w")
tot_busy = tot_gap = 0
for s, e, n in rows[idx:end + 1]:
    n = n.replace("void (anonymous namespace)::", "")[:70]
    out.write(f"{(s - rows[idx][0]) / 1e3:9.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {n}\n")
    tot_busy += e - s; tot_gap += max(0, s - prev); prev = max(prev, e)
out.write(f"launches {end + 1 - idx}  busy {tot_busy / 1e3:.1f} us  gaps {tot_gap / 1e3:.1f} us  span {(rows[end][1] - rows[idx][0]) / 1e3:.1f} us\n")
print(f"head: launches {end + 1 - idx}  busy {tot_busy / 1e3:.1f} us  gaps {tot_gap / 1e3:.1f} us  span {(rows[end][1] - rows[idx][0]) / 1e3:.1f} us")
PY
rm -rf gpurun_out/r06b_trace
