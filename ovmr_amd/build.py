"""Build libovmr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m ovmr_amd.build                 # incremental, parallel per translation unit
    python -m ovmr_amd.build --experiments   # libovmr_hip_exp.so: + A/B environment switches and timing-only ablation kernels
The library lands in ovmr_amd/lib/ (git-ignored, but it travels to the GPU box with gpurun).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libovmr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-mcode-object-version=5",
         "-Wno-unused-result", "-ffp-contract=off"]


def source_sha16(names=("gemm_f16_v5.hip", "common.h")) -> str:
    """sha256[:16] of kernel sources: profiles/*pmc*.json carry it so that bench.py refuses a counter summary taken on
    other code than the one it is timing."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose: bool = True, force: bool = False, experiments: bool = False) -> str:
    """experiments=True builds libovmr_hip_exp.so with -DOVMR_EXPERIMENTS: the environment A/B switches and the timing-only
    ablation kernels tools/gemm_bench.py uses (select it with OVMR_HIP_LIB); the product library has neither."""
    global OBJ, LIB
    if experiments:
        OBJ, LIB = os.path.join(HERE, "lib", "obj_exp"), os.path.join(HERE, "lib", "libovmr_hip_exp.so")
    else:
        OBJ, LIB = os.path.join(HERE, "lib", "obj"), os.path.join(HERE, "lib", "libovmr_hip.so")
    flags = FLAGS + (["-DOVMR_EXPERIMENTS"] if experiments else [])
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    if experiments:                                     # csrc/experiments/: kernels that exist for their measurements only (attention variants 4 and 6)
        srcs += sorted(os.path.join("experiments", f) for f in os.listdir(os.path.join(CSRC, "experiments")) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "ovmr_hip.h"))
    jobs = []
    for f in srcs:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, os.path.basename(f)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC, *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.basename(f)[:-4] + ".o") for f in srcs]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv))
