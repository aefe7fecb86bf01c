"""Build-time check of the hand-synchronised kernels' ISA (no GPU needed: hipcc cross-compiles).

Three kernels order LDS-DMA (`global_load_lds`) against LDS reads BY HAND -- a counted `s_waitcnt vmcnt(N)` and a raw `s_barrier`
at the top of their main loop -- and two of them issue the DMA from inline asm precisely so that hipcc does not track it
(attention_v3.hip / attention_v5.hip; the GEMM's ping-pong K loop uses the builtin and relies on hipcc adding no wait of its own).
Their correctness and their overlap both depend on what the compiler puts between the first and the last MFMA of the kernel:

  * every vector-memory wait in that region must be one of OURS, i.e. sit directly in front of a barrier (a wait hipcc adds for
    loads of its own -- it counts without the asm DMAs -- or for a spill reload would drain the prefetch ring in every iteration:
    that is how a first build of variant 5 lost 20 %);
  * no scratch traffic there (a spill reload is a VMEM op with exactly such a wait).

A compiler upgrade or an edit that breaks this fails here, on the CPU, instead of showing up as a slower or racy kernel."""
import os
import re
import subprocess

import pytest

from conftest import REPO

CSRC = os.path.join(REPO, "ovmr_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mcode-object-version=5", "-Wno-unused-result", "-ffp-contract=off",
         "-S", "--cuda-device-only", "-o", "-"]


def _kernels(src):
    """{mangled kernel name: [instruction lines]} of one translation unit."""
    r = subprocess.run([HIPCC, *FLAGS, os.path.join(CSRC, src)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur is not None:
            t = line.strip()
            if t.startswith("s_endpgm"):
                cur = None
            elif t and not t.startswith((";", ".", "#")) and not t.endswith(":"):
                out[cur].append(t)
    return out


def _check(name, ins, min_dma):
    mf = [i for i, t in enumerate(ins) if t.startswith("v_mfma")]
    assert mf, f"{name}: no MFMA found"
    body = ins[mf[0]:mf[-1] + 1]
    assert sum(t.startswith("global_load_lds") for t in ins) >= min_dma, f"{name}: LDS-DMA instructions missing"
    assert not any(t.startswith("scratch_") for t in body), f"{name}: spill traffic between the first and the last MFMA"
    bad = []
    for i, t in enumerate(body):
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            nxt = [u for u in body[i + 1:i + 12] if not u.startswith(("s_waitcnt", "s_nop", "s_mov", "s_cbranch", "s_branch", "s_cmp", "s_add",
                                                                       "s_and", "s_or", "s_lshl", "s_mul", "s_sub", "s_cselect", "s_andn2"))]
            if not (nxt and nxt[0].startswith("s_barrier")):
                bad.append((i, t, nxt[:2]))
    assert not bad, f"{name}: vector-memory waits in the main loop that are not followed by a barrier (compiler-added?): {bad[:4]}"


@pytest.mark.parametrize("src,pattern,min_dma", [
    ("attention_v5.hip", r"attn_f16_v5", 8),
    ("attention_v3.hip", r"attn_f16_v3", 2),
])
def test_attention_kernels_keep_only_the_hand_placed_waits(src, pattern, min_dma):
    ks = {k: v for k, v in _kernels(src).items() if re.search(pattern, k)}
    assert ks, f"no kernel matching {pattern} in {src}"
    for name, ins in ks.items():
        _check(name, ins, min_dma)


def test_gemm_ping_pong_loop_keeps_only_the_counted_waits():
    """The c_fc / in_proj / residual-projection instantiations of the 256-row tile kernel with the ping-pong K loop (OPT & 16)."""
    ks = _kernels("gemm_f16_v5.hip")
    picked = {k: v for k, v in ks.items() if re.search(r"gemm_f16_v5_kernelILi[367]ELi8ELi(17|528|2576)E", k)}
    assert len(picked) >= 3, sorted(ks)[:5]
    for name, ins in picked.items():
        mf = [i for i, t in enumerate(ins) if t.startswith("v_mfma")]
        body = ins[mf[0]:mf[-1] + 1]
        assert not any(t.startswith("scratch_") for t in body), f"{name}: spills in the K loop"
        waits = [t for t in body if t.startswith("s_waitcnt") and "vmcnt" in t]
        # ours: vmcnt(4) once per K-tile and vmcnt(0) for the last tile (the loop is rotated, so the first of them may sit in front
        # of the first MFMA); nothing else -- in particular no other count, which would be a wait hipcc computed for loads of its own
        assert waits and all(re.fullmatch(r"s_waitcnt vmcnt\((4|0)\)( lgkmcnt\(\d+\))?", w) for w in waits), f"{name}: {sorted(set(waits))}"
        assert any("vmcnt(4)" in w for w in waits), f"{name}: the counted wait is gone: {waits}"
        # the LDS-DMA of the loop is issued with scalar instructions only (scalar base + 32-bit lane offset, M0 from a scalar add):
        # no 64-bit vector address add and no v_readfirstlane per instruction beside the other row's MFMAs
        dma = [t for t in body if t.startswith("global_load_lds_dwordx4")]
        assert dma and all(re.match(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]", t) for t in dma), f"{name}: {sorted(set(dma))[:3]}"
        assert not any(t.startswith(("v_lshl_add_u64", "v_readfirstlane")) for t in body), f"{name}: vector address arithmetic in the K loop"
