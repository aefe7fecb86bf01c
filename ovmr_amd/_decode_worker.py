"""Decode worker of ovmr_amd.loader (a separate, torch-free module: worker processes are spawned and import only this).

The host's share of the reference's test transform (Dassl.pytorch/dassl/data/transforms/transforms.py:495-526) is now the DECODE
alone: open, RGB, and the raw uint8 [h, w, 3] frame goes into the shared upload ring; Resize + CenterCrop run on the GPU, bit-equal to
PIL (ovmr_resize_crop_u8), followed by ToTensor / Normalize / the fp16 cast (ovmr_preprocess_u8).  A frame larger than the ring's
per-image share, or an interpolation the device does not restate (nearest), is resized and cropped here with PIL as before
(`load_u8`) and travels as a finished crop."""
from __future__ import annotations

import numpy as np

DEVICE_INTERPOLATIONS = ("bicubic", "bilinear")


def _resample(name: str):
    from PIL import Image
    return {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST}[name]     # INTERPOLATION_MODES, transforms.py:20-24


def load_u8(src, size: int, fast: bool = False, interpolation: str = "bicubic") -> np.ndarray:
    """The whole host transform with PIL: resize of the smaller edge to `size`, centre crop -> uint8 [size, size, 3].
    src: a path (opened and closed here) or an already opened PIL image."""
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


def decode_into(path, arena: np.ndarray, offset: int, limit: int, size: int, cap: int, fast: bool, interpolation: str):
    """Decode `path` and place it at arena[offset:]: the raw RGB frame when it takes at most `cap` bytes (and the device restates the
    interpolation), else the finished size x size crop.  -> (bytes written, w, h, passthrough)."""
    from PIL import Image
    with Image.open(path) as img:
        if fast:
            img.draft("RGB", (size, size))
        w, h = img.size
        n = w * h * 3
        if interpolation in DEVICE_INTERPOLATIONS and n <= cap and offset + n <= limit:
            rgb = img.convert("RGB")
            w, h = rgb.size
            arena[offset:offset + n] = np.frombuffer(rgb.tobytes(), dtype=np.uint8)
            return n, w, h, 0
        crop = load_u8(img, size, False, interpolation)
    n = size * size * 3
    if offset + n > limit:
        raise RuntimeError("the upload ring's chunk is full")       # cannot happen: cap >= size * size * 3 per image
    arena[offset:offset + n] = crop.reshape(-1)
    return n, size, size, 1


def worker_main(shm_name: str, slot_bytes: int, slots: int, chunk_bytes: int, size: int, cap: int, fast: bool, task_q, done_q,
                interpolation: str = "bicubic") -> None:
    """Tasks: (slot, chunk index, [paths]); None stops the worker.
    Replies (slot, chunk index, [(byte offset inside the slot, w, h, passthrough), ...], error text or None)."""
    from multiprocessing import shared_memory
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        buf = np.ndarray((slots * slot_bytes,), dtype=np.uint8, buffer=shm.buf)
        while True:
            task = task_q.get()
            if task is None:
                break
            slot, ci, paths = task
            err, metas = None, []
            base = slot * slot_bytes
            off, limit = ci * chunk_bytes, (ci + 1) * chunk_bytes
            p = None
            try:
                for p in paths:
                    n, w, h, passthrough = decode_into(p, buf, base + off, base + limit, size, cap, fast, interpolation)
                    metas.append((off, w, h, passthrough))
                    off += (n + 15) // 16 * 16
            except Exception as e:          # noqa: BLE001 -- reported to the parent, which raises
                err = f"{type(e).__name__}: {e} ({p})"
            done_q.put((slot, ci, metas, err))
    finally:
        del buf
        shm.close()
