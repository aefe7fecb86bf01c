"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/ovmr_hip.h declares; the product package refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

from conftest import REPO


def _header_symbols():
    text = open(os.path.join(REPO, "include", "ovmr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ovmr_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from ovmr_amd import build, runtime
    if not os.path.exists(runtime.LIB_PATH):
        build.build(verbose=False)
    return runtime.load_library()


def test_header_symbols_are_exported(lib):
    from ovmr_amd import runtime
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ovmr_hip.h but not exported"
    assert set(syms) == set(runtime.SIGNATURES), "ctypes signature table and header disagree"


def test_version_and_argument_errors(lib):
    import ctypes
    from ovmr_amd import runtime
    assert b"gfx950" in lib.ovmr_version()
    assert lib.ovmr_create(None, None) == -1                       # OVMR_E_ARG, no compute, no GPU needed
    bad = runtime.ModelDesc(512, 224, 12, 700, 16, 77, 49408, 512, 12, 2, 4)   # width not a multiple of 64
    h = ctypes.c_void_p()
    assert lib.ovmr_create(ctypes.byref(bad), ctypes.byref(h)) == -2           # OVMR_E_SHAPE


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    from ovmr_amd import synth
    from ovmr_amd.runtime import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(synth.SPECS["tiny"])


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ovmr_amd/ may import or reference it."""
    for root, _, files in os.walk(os.path.join(REPO, "ovmr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_shard_helpers():
    from ovmr_amd.shard import shard_batches, shard_range
    for n in (1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            got = sorted(sum((shard_batches(n, r, w) for r in range(w)), []))
            assert got == list(range(n))
