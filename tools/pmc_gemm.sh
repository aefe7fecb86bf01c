#!/bin/bash
# PMC passes over the GEMM micro-benchmark (one variant), as MI355X_MICROARCH.md prescribes: counters in
# their own runs, --kernel-trace only.  usage: tools/pmc_gemm.sh <variant> <outdir>
V=${1:-4}; OUT=${2:-gpurun_out/pmc}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/gemm_bench.py --variants $V --rounds 1 --reps 2 > $R/$OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
cd $R
python3 - <<PY
import csv,glob,collections
for name in ("fetch","write","sq1","sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%name, recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "gemm" not in k: continue
            key=(k[-40:], r.get("Grid_Size",""))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for key,c in agg.items():
            print(name, key, {n: round(sum(v)/len(v),1) for n,v in c.items()})
PY
