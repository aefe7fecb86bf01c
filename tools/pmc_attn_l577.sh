#!/bin/bash
OUT=${1:-gpurun_out/pmc_attn_l577}; R=$(pwd)
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/attn_bench.py --only vit-l336 --reps 2 --variants 1 5 > $R/$OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU
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
            if "attn_f16" not in k:
                continue
            key = k.split("::")[-1].split("(")[0]
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[key]["duration_us_under_pmc"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        for key, c in agg.items():
            for n, v in c.items():
                out[key][n] = round(sum(v) / len(v), 2)
for key, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, c in out.items():
    print(k, json.dumps(c))
PY
