#!/usr/bin/env python3
"""Per launch shape, the dominant kernel's durations in a rocprofv3 kernel trace of `bench.py` beside the figures of the bench line's
roofline probe (launches running ALONE).  `rocprofv3 --stats` averages by kernel NAME: since the inference phase keeps two batches in
flight (CustomCLIP.forward_batches) its M = 50 432 launches overlap in pairs and each one's wall duration in the trace is about twice its
share of the machine, so the name-wide average no longer equals the probe's launch-weighted one; per grid the generation launches do.

    python tools/dominant_by_grid.py <kernel_trace.csv> <bench line .json> [out.json]
"""
import collections, csv, json, sys

trace, line = sys.argv[1], sys.argv[2]
d = json.load(open(line))
name = "gemm_f16_v5_kernel<7, 8, 2576>"                  # c_fc: ln_2 fold + bias + one-rounding QuickGELU, 256-row tiles, ping-pong K loop
agg = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    if name in r["Kernel_Name"]:
        agg[int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
probe = {((s["M"] + 255) // 256) * 12: s for s in d["roofline"]["shapes"]}      # N = 3072: 12 column tiles
out = {"kernel": name, "in_flight_during_inference": d["phases"].get("inference_batches_in_flight"), "by_grid": []}
for g, v in sorted(agg.items()):
    v.sort()
    p = probe.get(g)
    out["by_grid"].append({"workgroups": g, "M": p["M"] if p else None, "calls": len(v), "trace_avg_us": round(sum(v) / len(v), 1),
                           "trace_median_us": round(v[len(v) // 2], 1), "probe_avg_us_alone": p["avg_launch_us"] if p else None})
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
