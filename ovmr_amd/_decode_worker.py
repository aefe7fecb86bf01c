"""Decode worker of ovmr_amd.loader (a separate, torch-free module: worker processes are spawned and import only this).

The host's share of the reference's test transform (Dassl.pytorch/dassl/data/transforms/transforms.py:495-526): open, RGB,
resize of the smaller edge to `size` (INPUT.INTERPOLATION; bicubic in the MM_CLS_OP configs), centre crop -- uint8 [size, size, 3], written straight into the shared batch buffer.
ToTensor / Normalize / the fp16 cast run on the GPU (ovmr_preprocess_u8)."""
from __future__ import annotations

import numpy as np


def _resample(name: str):
    from PIL import Image
    return {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST}[name]     # INTERPOLATION_MODES, transforms.py:20-24


def load_u8(src, size: int, fast: bool = False, interpolation: str = "bicubic") -> np.ndarray:
    """src: a path (opened and closed here) or an already opened PIL image."""
    from PIL import Image
    if isinstance(src, (str, bytes)) or hasattr(src, "__fspath__"):
        with Image.open(src) as img:
            return load_u8(img, size, fast, interpolation)
    img = src
    if fast:
        img.draft("RGB", (size, size))          # JPEG only: DCT-domain downscale to >= size (changes pixels slightly; off by default)
    img = img.convert("RGB")
    w, h = img.size
    if w <= h:
        nw, nh = size, max(size, int(size * h / w))
    else:
        nw, nh = max(size, int(size * w / h)), size
    img = img.resize((nw, nh), _resample(interpolation))
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    return np.asarray(img.crop((left, top, left + size, top + size)), dtype=np.uint8)


def worker_main(shm_name: str, slots: int, batch: int, size: int, fast: bool, task_q, done_q, interpolation: str = "bicubic") -> None:
    """Tasks: (slot, first index in the batch, [paths]); None stops the worker.  Replies (slot, count, error text or None)."""
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        buf = np.ndarray((slots, batch, size, size, 3), dtype=np.uint8, buffer=shm.buf)
        while True:
            task = task_q.get()
            if task is None:
                break
            slot, first, paths = task
            err = None
            try:
                for k, p in enumerate(paths):
                    buf[slot, first + k] = load_u8(p, size, fast, interpolation)
            except Exception as e:          # noqa: BLE001 -- reported to the parent, which raises
                err = f"{type(e).__name__}: {e} ({p})"
            done_q.put((slot, len(paths), err))
    finally:
        del buf
        shm.close()
