#!/bin/bash
# PMC passes over the GEMM micro-benchmark (one variant), as MI355X_MICROARCH.md prescribes: counters in their own
# runs, --kernel-trace only (never with --stats / sys-trace).  Writes <outdir>/summary.json with per-kernel means.
# usage: tools/pmc_gemm.sh <variant> <outdir> [batch]
V=${1:-6}; OUT=${2:-gpurun_out/pmc}; B=${3:-512}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/gemm_bench.py --variants $V --rounds 1 --reps 2 --batch $B > $R/$OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
cd $R
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
for name in ("fetch", "write", "sq1", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm" not in k:
                continue
            key = k.split("::")[-1].split("(")[0] + " grid=" + r["Grid_Size"]
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[key]["duration_us_under_pmc"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        for key, c in agg.items():
            for n, v in c.items():
                out[key][n] = round(sum(v) / len(v), 2)
for key, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # gfx950: FETCH_SIZE (KiB) reports half of the bytes of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM)
        c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
json.dump({"variant": $V, "batch": $B, "kernels": out}, open("$OUT/summary.json", "w"), indent=1)
for k, c in out.items():
    print(k, c)
PY
