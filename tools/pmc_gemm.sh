#!/bin/bash
# PMC passes over the GEMM micro-benchmark (one variant, PRODUCT library), as MI355X_MICROARCH.md prescribes: counters in their
# own runs, --kernel-trace only (never with --stats / sys-trace).  Writes <outdir>/summary.json with per-kernel means, the derived
# figures bench.py puts in its roofline object (mfma_busy_frac, hbm_gbps, lds_conflict_frac, clock_ghz) and the hash of the
# kernel sources they were taken on (ovmr_amd.build.source_sha16: bench.py refuses a summary of other code).
# usage: tools/pmc_gemm.sh <variant> <outdir> [batch]
V=${1:-6}; OUT=${2:-gpurun_out/pmc}; B=${3:-512}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
# SHAPES: "" = every shape of gemm_bench.py in one process (kernels keyed by name + grid); a shape name = that shape alone, keyed
# "... shape=<name>": out_proj (K = 768) and c_proj (K = 3072) share one instantiation <3, 8, 17> AND one grid, so only separate
# passes tell them apart (r04).
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name$TAG -- python3 $R/tools/gemm_bench.py --variants $V --rounds 1 --reps 2 --batch $B --shapes "$SHAPE" > $R/$OUT/$name$TAG.log 2>&1; }
for SHAPE in "" ${PMC_SHAPES:-out_proj c_proj out_proj_st c_proj_st}; do
TAG=${SHAPE:+_$SHAPE}
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
done
cd $R
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
from ovmr_amd.build import source_sha16
out = collections.defaultdict(dict)
import os
shapes = [""] + "${PMC_SHAPES:-out_proj c_proj out_proj_st c_proj_st}".split()
for name, shape in [(n, sh) for sh in shapes for n in ("fetch", "write", "sq1", "sq2")]:
    for f in glob.glob("$OUT/%s%s/**/*counter_collection.csv" % (name, "_" + shape if shape else ""), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm" not in k:
                continue
            key = k.split("::")[-1].split("(")[0] + " grid=" + r["Grid_Size"] + (" shape=" + shape if shape else "")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[key]["duration_us_under_pmc"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        for key, c in agg.items():
            for n, v in c.items():
                out[key][n] = round(sum(v) / len(v), 2)
for key, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # gfx950: FETCH_SIZE (KiB) reports half of the bytes of wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM)
        c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        c["hbm_gbps"] = round(c["hbm_bytes_per_launch"] / (c["duration_us_under_pmc"] * 1e-6) / 1e9, 1)
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                        # the counter sums the 8 XCDs (MI355X_MICROARCH.md, DVFS)
        c["clock_ghz"] = round(cycles / (c["duration_us_under_pmc"] * 1e3), 3)
        c["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024), 4)   # 256 CUs x 4 matrix pipes
    if c.get("SQ_LDS_IDX_ACTIVE"):
        c["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
json.dump({"variant": $V, "batch": $B, "kernel_source_sha16": source_sha16(), "kernels": out}, open("$OUT/summary.json", "w"), indent=1)
for k, c in out.items():
    print(k, c)
PY
