#!/bin/bash
# PMC passes over the classifier head's kernels (HBM GB/s and MFMA utilisation "for both" -- the encoder AND the head): the
# one-launch inference head (tools/head_bench.py), the cross-validation step (tools/xval_bench.py: logits GEMM with the fused argmax epilogue +
# xval_argmax_reduce) and the device transform (tools/resize_sweep.py).  Counters in their own runs, --kernel-trace only (MI355X_MICROARCH.md).
# usage: tools/pmc_head.sh <outdir>
OUT=${1:-gpurun_out/pmc_head}; R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT; cd /tmp; export TMPDIR=/tmp
run() { name=$1; prog=$2; shift 2; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/$OUT/$name -- python3 $R/tools/$prog > $R/$OUT/$name.log 2>&1; }
for prog in "head_bench.py" "xval_bench.py --classes 10000 --shots 8 --reps 1" "resize_sweep.py --n 200"; do
  tag=$(echo $prog | cut -d. -f1)
  run ${tag}_fetch "$prog" FETCH_SIZE
  run ${tag}_write "$prog" WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  run ${tag}_sq "$prog" SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
done
cd $R
python3 - <<PY
import csv, glob, collections, json
out = collections.defaultdict(dict)
keep = ("head_fused", "xval_argmax", "gemm_f16_v5_kernel<8", "fused_softmax", "resize_h", "resize_v", "preprocess_u8", "scale_f16")
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in keep):
            continue
        key = k.split("::")[-1].split("(")[0] + " grid=" + r["Grid_Size"]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["duration_us_under_pmc"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        agg[key]["launches"].append(1.0)
    for key, c in agg.items():
        for n, v in c.items():
            out[key][n] = round(sum(v), 0) if n == "launches" else round(sum(v) / len(v), 2)
for key, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        c["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024       # gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)
        c["hbm_gbps"] = round(c["hbm_bytes_per_launch"] / (c["duration_us_under_pmc"] * 1e-6) / 1e9, 1)
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and c["GRBM_GUI_ACTIVE"] > 0:
        c["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024), 4)
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, c in sorted(out.items()):
    print(k[:90], {x: c[x] for x in ("duration_us_under_pmc", "hbm_bytes_per_launch", "hbm_gbps", "mfma_busy_frac") if x in c})
PY
