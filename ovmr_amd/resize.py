"""Coefficient tables of PIL's two-pass fixed-point resize, for the device half of the test transform (ovmr_resize_crop_u8).

The reference's test transform is torchvision `Resize(max(INPUT.SIZE), interpolation)` + `CenterCrop(INPUT.SIZE)`
(Dassl.pytorch/dassl/data/transforms/transforms.py:495-526) on PIL images, i.e. `PIL.Image.resize` -- Pillow, a third-party
dependency that is not vendored in the reference (pinned through torchvision 0.15.2, README.md:31; here Pillow 12).  Its
algorithm for 8-bit images (src/libImaging/Resample.c; unchanged since Pillow 3.4) is restated here:

  * `precompute_coeffs`: per output pixel a window [xmin, xmin + n) of input pixels and double-precision filter weights
    filter((x + xmin - center + 0.5) / filterscale), filterscale = max(scale, 1), support = filter.support * filterscale,
    center = (xx + 0.5) * scale, normalised to sum 1;
  * `normalize_coeffs_8bpc`: weights rounded half away from zero to 22-bit fixed point (PRECISION_BITS = 32 - 8 - 2);
  * `ImagingResampleHorizontal_8bpc` then `ImagingResampleVertical_8bpc`: ss = 2^21 + sum pixel * weight in int32,
    clip8(ss >> 22), the intermediate image is uint8.

Only the host side lives here: the integer tables for the R x R crop window.  The convolution itself is integer arithmetic on
the GPU (csrc/resize_crop.hip), so the result equals PIL's BIT FOR BIT (tests/test_hip_loader.py holds it to `load_u8` on 200+
sizes).  Nearest-neighbour resizing is another algorithm in PIL and stays on the host.
"""
from __future__ import annotations

from functools import lru_cache
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2

_SUPPORT = {"bicubic": 2.0, "bilinear": 1.0}


def _filter(name: str, x: np.ndarray) -> np.ndarray:
    x = np.abs(x)
    if name == "bilinear":                              # bilinear_filter
        return np.where(x < 1.0, 1.0 - x, 0.0)
    a = -0.5                                            # bicubic_filter (Keys, a = -0.5)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def _coeffs(in_size: int, out_size: int, first: int, count: int, name: str) -> Tuple[np.ndarray, np.ndarray]:
    """precompute_coeffs + normalize_coeffs_8bpc for output pixels [first, first + count) of an axis resized in_size -> out_size
    (box = the whole axis).  Returns bounds int32 [count, 2] = (xmin, n) and weights int32 [count, ksize]."""
    scale = in_size / out_size                          # (in1 - in0) / outSize, doubles as in C
    filterscale = max(scale, 1.0)
    support = _SUPPORT[name] * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (first + np.arange(count, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)        # C's (int) truncates; negative values are clamped at 0 right after
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    j = np.arange(ksize, dtype=np.int64)[None, :]
    live = j < n[:, None]
    w = np.where(live, _filter(name, (j + xmin[:, None] - center[:, None] + 0.5) * ss), 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]                   # the C loop's left-to-right summation (the padding adds exact zeros)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    # normalize_coeffs_8bpc: (int)(k * 2^22 -+ 0.5), truncation toward zero
    fixed = np.where(w < 0, np.trunc(-0.5 + w * (1 << PRECISION_BITS)), np.trunc(0.5 + w * (1 << PRECISION_BITS)))
    kk = np.where(live, fixed, 0.0).astype(np.int64).astype(np.int32)
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    return bounds, kk


def resized_size(w: int, h: int, size: int) -> Tuple[int, int]:
    """torchvision.transforms.Resize(size) with an int: the smaller edge becomes `size`, the other int(size * long / short)."""
    if w <= h:
        return size, max(size, int(size * h / w))
    return max(size, int(size * w / h)), size


@lru_cache(maxsize=4096)
def plan(w: int, h: int, size: int, interpolation: str = "bicubic"):
    """Tables for one input size: Resize(size) + CenterCrop(size) of a w x h image.
    -> dict(y0, ny: the input rows the crop window needs; table: int32 array laid out as
       [bounds_h (size x 2) | k_h (size x ksize_h) | bounds_v (size x 2, ymin relative to y0) | k_v (size x ksize_v)],
       ksize_h, ksize_v)."""
    if interpolation not in _SUPPORT:
        raise ValueError(f"no device resize for interpolation '{interpolation}'")
    nw, nh = resized_size(w, h, size)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))     # torchvision center_crop
    bh, kh = _coeffs(w, nw, left, size, interpolation)
    bv, kv = _coeffs(h, nh, top, size, interpolation)
    y0 = int(bv[:, 0].min())
    y1 = int((bv[:, 0] + bv[:, 1]).max())
    bv = bv.copy()
    bv[:, 0] -= y0
    table = np.concatenate([bh.reshape(-1), kh.reshape(-1), bv.reshape(-1), kv.reshape(-1)]).astype(np.int32)
    return {"y0": y0, "ny": y1 - y0, "table": table, "ksize_h": kh.shape[1], "ksize_v": kv.shape[1]}


def resize_crop_reference(img: np.ndarray, size: int, interpolation: str = "bicubic") -> np.ndarray:
    """The two integer passes in numpy (uint8 [h, w, 3] -> [size, size, 3]): what the GPU kernel computes.  CPU tests hold THIS to PIL;
    GPU tests hold the kernel to PIL directly."""
    h, w = img.shape[:2]
    p = plan(w, h, size, interpolation)
    t = p["table"]
    kh_n, kv_n = p["ksize_h"], p["ksize_v"]
    o = 0
    bh = t[o:o + 2 * size].reshape(size, 2); o += 2 * size
    kh = t[o:o + size * kh_n].reshape(size, kh_n); o += size * kh_n
    bv = t[o:o + 2 * size].reshape(size, 2); o += 2 * size
    kv = t[o:o + size * kv_n].reshape(size, kv_n)
    rows = img[p["y0"]:p["y0"] + p["ny"]].astype(np.int64)
    tmp = np.zeros((p["ny"], size, 3), dtype=np.uint8)
    for x in range(size):
        x0, n = bh[x]
        acc = (1 << (PRECISION_BITS - 1)) + (rows[:, x0:x0 + n] * kh[x, :n, None].astype(np.int64)).sum(1)
        tmp[:, x] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((size, size, 3), dtype=np.uint8)
    t64 = tmp.astype(np.int64)
    for y in range(size):
        y_0, n = bv[y]
        acc = (1 << (PRECISION_BITS - 1)) + (t64[y_0:y_0 + n] * kv[y, :n, None, None].astype(np.int64)).sum(0)
        out[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out
