#!/bin/bash
# PMC passes over tools/fused_qkv_bench.py at 775 images (experiment build): today's in_proj (gemm_f16_v5 <6, 8, ...>) and attention (attn_f16_v3)
# against the fused kernel (qkv_attn_fused_kernel) -- matrix-pipe and LDS counters, HBM bytes per launch.  Counters in their own runs,
# --kernel-trace only.   usage: tools/pmc_fused.sh <outdir>
OUT=${1:-gpurun_out/pmc_fused}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/fused_qkv_bench.py --batches 775 --reps 2 > $R/$OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $R
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for name in ("sq1", "sq2", "sq3", "fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            key = "fused" if "qkv_attn_fused" in k else "attention_v3" if "attn_f16_v3" in k else "in_proj" if ("gemm_f16_v5_kernel<6" in k) else None
            if key is None:
                continue
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[key]["duration_us_under_pmc"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        for key, c in agg.items():
            for n, v in c.items():
                out[key][n] = round(sum(v) / len(v), 2)
for key, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024      # gfx950: FETCH_SIZE (KiB) reports half of wide coalesced reads
        c["hbm_gbps"] = round(c["hbm_bytes_per_launch"] / (c["duration_us_under_pmc"] * 1e-6) / 1e9, 1)
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        c["clock_ghz"] = round(cycles / (c["duration_us_under_pmc"] * 1e3), 3)
        c["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024), 4)
        if "SQ_LDS_IDX_ACTIVE" in c:
            c["lds_idx_active_frac_of_cu_cycles"] = round(c["SQ_LDS_IDX_ACTIVE"] / (cycles * 256), 4)
            c["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/*/
