"""Pipelined folder loader: the input side of `ovmr_amd.cli`.

The reference's loaders run 8 DataLoader workers with pinned memory (Dassl.pytorch/dassl/data/data_manager.py:69-113,
configs/trainers/MM_CLS_OP/vit_b16_c4_ep50_imagenet21k_pretrain.yaml:9) and hand fp32 CHW tensors to the model, which moves them
to the GPU and casts to fp16.  Here the split is:

  worker processes (N, spawned, torch-free)   DECODE only: the raw uint8 RGB frame goes into a shared, page-locked upload ring (PIL's
                                              bicubic resize was ~60 % of a worker's time per image); the pool and the ring outlive a
                                              pass (exemplar set, then test set: one start-up)
  side HIP stream                             asynchronous H2D copies of a finished batch's chunks, then ovmr_resize_crop_u8 (Resize +
                                              CenterCrop, PIL's own integer passes: bit-equal) and ovmr_preprocess_u8 (uint8 HWC ->
                                              normalised fp16 CHW: ToTensor + Normalize + .half() of the reference, bit for bit)
  compute stream (the caller's)               waits on the batch's event, runs the encoder; the workers are already decoding
                                              `prefetch` batches ahead, the side stream uploads batch i+1 under the encoder of batch i

It yields the loader protocol's dict batches {"img": fp16 [B, 3, R, R] on the device, "label": int64 [B]}."""
from __future__ import annotations

import ctypes
import time
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)      # configs/trainers/MM_CLS_OP/*.yaml:14-15
PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_u8(u8: torch.Tensor, out: torch.Tensor = None, stream=None, mean=PIXEL_MEAN, std=PIXEL_STD) -> torch.Tensor:
    """uint8 [B, R, R, 3] on the device -> normalised fp16 [B, 3, R, R] (ovmr_preprocess_u8).  mean=None: ToTensor only (no Normalize)."""
    if mean is None:
        mean, std = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    from . import runtime
    lib = runtime.load_library()
    assert u8.is_cuda and u8.dtype == torch.uint8 and u8.dim() == 4 and u8.shape[3] == 3 and u8.is_contiguous()
    B, R = int(u8.shape[0]), int(u8.shape[1])
    if out is None:
        out = torch.empty((B, 3, R, R), dtype=torch.float16, device=u8.device)
    mean, std = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    s = stream if stream is not None else torch.cuda.current_stream(u8.device)
    rc = lib.ovmr_preprocess_u8(ctypes.c_void_p(u8.data_ptr()), B, R, mean, std, ctypes.c_void_p(out.data_ptr()),
                                ctypes.c_void_p(s.cuda_stream))
    if rc != 0:
        raise runtime.OvmrError(f"ovmr_preprocess_u8 failed with {rc}")
    return out


def resize_crop_u8(frames, size: int, interpolation: str = "bicubic", device="cuda:0", stream=None) -> torch.Tensor:
    """Resize(size) + CenterCrop(size) of decoded frames on the GPU, bit-equal to PIL (ovmr_resize_crop_u8).
    frames: list of uint8 arrays [h, w, 3] (any sizes).  -> uint8 [n, size, size, 3] on the device."""
    r = DeviceResizer(size, interpolation, device)
    offs, metas, off = [], [], 0
    for f in frames:
        f = np.ascontiguousarray(f, dtype=np.uint8)
        assert f.ndim == 3 and f.shape[2] == 3
        metas.append((off, f.shape[1], f.shape[0], 0))
        offs.append((off, f))
        off += (f.size + 15) // 16 * 16
    arena = np.zeros(max(off, 16), dtype=np.uint8)
    for o, f in offs:
        arena[o:o + f.size] = f.reshape(-1)
    out = torch.empty((len(frames), size, size, 3), dtype=torch.uint8, device=device)
    s = stream if stream is not None else torch.cuda.current_stream(torch.device(device))
    with torch.cuda.stream(s):
        pixels = torch.from_numpy(arena).to(device)
        r.run(pixels, metas, out, s)
    return out


class DeviceResizer:
    """Builds the jobs and coefficient tables of a batch (ovmr_amd.resize.plan, one table per distinct input size) and launches
    ovmr_resize_crop_u8 on a stream.  The device buffers for jobs, tables and the horizontal pass's intermediate are kept between
    batches and grown on demand (allocated on the stream they are used on)."""

    def __init__(self, size: int, interpolation: str = "bicubic", device="cuda:0"):
        from . import runtime
        self.lib = runtime.load_library()
        self.Job = runtime.ResizeJob
        self.size, self.interpolation, self.device = int(size), interpolation, torch.device(device)
        self._tmp = self._tables = self._jobs = None

    def _buf(self, name: str, nbytes: int, dtype) -> torch.Tensor:
        cur = getattr(self, name)
        item = torch.empty((), dtype=dtype).element_size()
        n = (nbytes + item - 1) // item
        if cur is None or cur.numel() < n:
            cur = torch.empty((max(n, 1) * 5 // 4 + 16,), dtype=dtype, device=self.device)
            setattr(self, name, cur)
        return cur

    def run(self, pixels: torch.Tensor, metas, out_u8: torch.Tensor, stream) -> None:
        """pixels: device uint8 arena; metas: [(byte offset in the arena, w, h, passthrough)] per image; out_u8: device uint8
        [n, size, size, 3].  Everything is enqueued on `stream`."""
        from . import resize
        n, R = len(metas), self.size
        if n == 0:
            return
        jobs = (self.Job * n)()
        tables, where, t_off, tmp_off, max_ny = [], {}, 0, 0, 0
        for j, (off, w, h, passthrough) in zip(jobs, metas):
            j.in_offset, j.w, j.h, j.passthrough = int(off), int(w), int(h), int(passthrough)
            if passthrough:
                continue
            key = (int(w), int(h))
            if key not in where:
                p = resize.plan(key[0], key[1], R, self.interpolation)
                where[key] = (t_off, p)
                tables.append(p["table"])
                t_off += p["table"].size
            j.table, p = where[key]
            j.y0, j.ny, j.ksize_h, j.ksize_v = p["y0"], p["ny"], p["ksize_h"], p["ksize_v"]
            j.tmp_offset = tmp_off
            tmp_off += (p["ny"] * R * 3 + 15) // 16 * 16
            max_ny = max(max_ny, p["ny"])
        with torch.cuda.stream(stream):
            tab = self._buf("_tables", max(t_off, 1) * 4, torch.int32)
            if tables:
                tab[:t_off].copy_(torch.from_numpy(np.concatenate(tables)), non_blocking=True)
            jb = self._buf("_jobs", ctypes.sizeof(jobs), torch.uint8)
            jb[:ctypes.sizeof(jobs)].copy_(torch.frombuffer(bytearray(jobs), dtype=torch.uint8), non_blocking=True)
            tmp = self._buf("_tmp", max(tmp_off, 16), torch.uint8)
            rc = self.lib.ovmr_resize_crop_u8(ctypes.c_void_p(pixels.data_ptr()), ctypes.c_void_p(jb.data_ptr()), n,
                                              ctypes.c_void_p(tab.data_ptr()), ctypes.c_void_p(tmp.data_ptr()), int(max_ny),
                                              ctypes.c_void_p(out_u8.data_ptr()), R, ctypes.c_void_p(stream.cuda_stream))
        if rc != 0:
            from . import runtime
            raise runtime.OvmrError(f"ovmr_resize_crop_u8 failed with {rc}")


def _shm_free_bytes():
    """Free bytes of the tmpfs behind multiprocessing.shared_memory (None where there is no /dev/shm to ask)."""
    import shutil
    try:
        return int(shutil.disk_usage("/dev/shm").free)
    except OSError:
        return None


class _Pool:
    """Decode workers + the shared, page-locked upload ring.  Kept alive between passes (module cache, torn down at exit or by
    close_pools()): spawning 8-16 interpreters and registering the ring costs ~0.3-0.5 s, a third of a 1000-image pass."""

    def __init__(self, key):
        import multiprocessing as mp
        from multiprocessing import shared_memory
        from . import _decode_worker
        (self.workers, self.slots, self.B, self.R, self.cap, self.chunk, self.interpolation, self.fast) = self.key = key
        self.n_chunks = (self.B + self.chunk - 1) // self.chunk
        self.chunk_bytes = self.chunk * ((self.cap + 15) // 16 * 16)
        self.slot_bytes = self.n_chunks * self.chunk_bytes
        nbytes = self.slots * self.slot_bytes
        # SharedMemory only ftruncate()s, which tmpfs grants beyond its free space: a ring larger than /dev/shm would be found out by the
        # workers, as SIGBUS on their first store into a page that cannot be backed (Docker's default /dev/shm is 64 MiB; this ring is
        # 576 MiB at batch 256 x 3 slots x 768 KiB).  Ask first -- the caller (_pool) then halves the share per image.
        free = _shm_free_bytes()
        if free is not None and nbytes + (16 << 20) > free:
            raise OSError(f"the upload ring needs {nbytes >> 20} MiB of /dev/shm, {free >> 20} MiB are free")
        self.shm = shared_memory.SharedMemory(create=True, size=nbytes)
        self.host = torch.from_numpy(np.ndarray((self.slots, self.slot_bytes), dtype=np.uint8, buffer=self.shm.buf))
        self.cudart = torch.cuda.cudart()
        self.pinned = int(self.cudart.cudaHostRegister(self.host.data_ptr(), nbytes, 0)) == 0     # DMA straight out of the shared ring
        ctx = mp.get_context("spawn")               # this process has initialised the GPU: never fork it
        self.task_q, self.done_q = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_decode_worker.worker_main,
                                  args=(self.shm.name, self.slot_bytes, self.slots, self.chunk_bytes, self.R, self.cap, self.fast,
                                        self.task_q, self.done_q, self.interpolation), daemon=True) for _ in range(self.workers)]
        for p in self.procs:
            p.start()
        self.busy = False

    def close(self):
        for _ in self.procs:
            try:
                self.task_q.put(None)
            except Exception:                        # noqa: BLE001
                pass
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.terminate()
        try:
            if self.pinned:
                torch.cuda.synchronize()
                self.cudart.cudaHostUnregister(self.host.data_ptr())
        except Exception:                            # noqa: BLE001
            pass
        self.host = None
        try:
            self.shm.close()
            self.shm.unlink()
        except Exception:                            # noqa: BLE001
            pass


_POOLS: Dict[tuple, _Pool] = {}


def close_pools() -> None:
    """Stop every cached decode pool and release its ring."""
    for k in list(_POOLS):
        _POOLS.pop(k).close()


def _get_pool(key) -> _Pool:
    """A cached pool with the same workers / ring depth / image size / share per image / chunk / transform whose slots hold at least
    this batch size, else a new one."""
    import atexit
    p = None
    for k in list(_POOLS):
        q = _POOLS[k]
        if q.busy or any(not w.is_alive() for w in q.procs):
            _POOLS.pop(k).close()                    # an abandoned pass left tasks behind, or a worker died
        elif k[:2] == key[:2] and k[3:] == key[3:] and k[2] >= key[2]:
            p = q
    if p is None:
        if not _POOLS:
            atexit.register(close_pools)
        for k in list(_POOLS):                       # one ring at a time (hundreds of MB of page-locked memory each)
            _POOLS.pop(k).close()
        p = _POOLS[key] = _Pool(key)
    return p


class PipelinedFolderLoader:
    """Iterable of {"img", "label"} dict batches over (path, label) items; see the module docstring.

    With world > 1 the loader is class-sharded like cli.FolderLoader (`presharded = True`).  `stats` after an iteration:
    images, batches, wall_s, decode_wait_s (the caller blocked on the workers), decode_bound_fraction = decode_wait_s / wall_s (the
    share of the job the host spent waiting for JPEG decode: robust whatever streams the consumer uses), consumer_gpu_s (device time
    on the caller's CURRENT stream between a batch becoming available and the caller asking for the next one) and
    encoder_idle_fraction = 1 - consumer_gpu_s / wall_s -- an UPPER bound when the consumer runs the encoder on side streams
    (CustomCLIP.forward_batches: only the staging copy and the hand-over wait fall on the current stream); device_resized /
    host_resized: images whose Resize + CenterCrop ran on the GPU / in the worker (frames above `raw_cap_bytes`, nearest).

    raw_cap_bytes: the upload ring's share per image (default 768 KiB: a 500 x 500 frame; ImageNet's typical 500 x 375 takes 549 KiB);
    device_resize=False restores the host transform for every image (the ring then holds finished crops only)."""

    def __init__(self, items: Sequence[Tuple[str, int]], batch_size: int, size: int, rank: int = 0, world: int = 1,
                 num_classes: int = 0, workers: int = 8, prefetch: int = 3, device: str = "cuda:0", chunk: int = 8,
                 fast_decode: bool = False, interpolation: str = "bicubic", mean=PIXEL_MEAN, std=PIXEL_STD,
                 device_resize: bool = True, raw_cap_bytes: int = 768 * 1024):
        self.bs, self.size = int(batch_size), int(size)
        self.interpolation, self.mean, self.std = interpolation, mean, std
        self.presharded = world > 1
        if world > 1:
            from .shard import shard_range
            lo, hi = shard_range(num_classes or (1 + max(l for _, l in items)), rank, world)
            items = [it for it in items if lo <= it[1] < hi]
        self.items = list(items)
        self.workers, self.prefetch, self.chunk, self.fast = max(1, int(workers)), max(2, int(prefetch)), max(1, int(chunk)), fast_decode
        self.device = torch.device(device)
        crop = self.size * self.size * 3
        self.cap = max(crop, int(raw_cap_bytes)) if device_resize else crop
        self.stats: Dict[str, float] = {}

    def __len__(self):
        return (len(self.items) + self.bs - 1) // self.bs

    def warm(self) -> "PipelinedFolderLoader":
        """Start (or find) the decode pool now: the worker interpreters come up while the caller is still loading weights."""
        if len(self.items):
            self._pool()
        return self

    def _pool(self) -> _Pool:
        cap = self.cap
        while True:
            try:
                return _get_pool((self.workers, self.prefetch, self.bs, self.size, cap, self.chunk, self.interpolation, self.fast))
            except OSError as e:                     # /dev/shm too small for this ring: a smaller share per image (more host resizes)
                crop = self.size * self.size * 3
                if cap <= crop:
                    raise OSError(f"{e}; even one {self.size} x {self.size} crop per image does not fit: use a smaller TEST.BATCH_SIZE, "
                                  "--prefetch 2, --workers 0 (no ring), or a larger /dev/shm") from None
                cap = max(crop, cap // 2)

    def __iter__(self):
        nb = len(self)
        if nb == 0:
            return
        pool = self._pool()
        pool.busy = True
        B, R, slots = self.bs, self.size, pool.slots
        host, task_q, done_q, procs, pinned = pool.host, pool.task_q, pool.done_q, pool.procs, pool.pinned
        n_chunks, chunk_bytes = pool.n_chunks, pool.chunk_bytes
        staging = None if pinned else torch.empty((chunk_bytes,), dtype=torch.uint8).pin_memory()
        side = torch.cuda.Stream(self.device)
        resizer = DeviceResizer(R, self.interpolation if self.interpolation in ("bicubic", "bilinear") else "bicubic", self.device)
        dev_px = [torch.empty((pool.slot_bytes,), dtype=torch.uint8, device=self.device) for _ in range(2)]
        dev_u8 = [torch.empty((B, R, R, 3), dtype=torch.uint8, device=self.device) for _ in range(2)]
        dev_f16 = [torch.empty((B, 3, R, R), dtype=torch.float16, device=self.device) for _ in range(2)]
        pending = [0] * slots                       # chunks of the slot still being decoded
        metas = [dict() for _ in range(slots)]      # chunk index -> [(offset, w, h, passthrough)]
        copied = [None] * slots                     # event: the H2D copies out of a ring slot have finished (the slot may be refilled)
        consumed = [None, None]                     # event on the compute stream: the caller is done with device buffer k
        t_wait = gpu_ms = 0.0
        n_dev = n_host = 0
        timers: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []

        def submit(b: int) -> None:
            slot = b % slots
            if copied[slot] is not None:
                copied[slot].synchronize()
            chunk_items = self.items[b * B:(b + 1) * B]
            metas[slot] = {}
            pending[slot] = (len(chunk_items) + self.chunk - 1) // self.chunk
            for ci, s in enumerate(range(0, len(chunk_items), self.chunk)):
                task_q.put((slot, ci, [p for p, _ in chunk_items[s:s + self.chunk]]))

        t0 = time.perf_counter()
        ok = False
        try:
            for b in range(min(slots, nb)):
                submit(b)
            for b in range(nb):
                slot, k = b % slots, b % 2
                compute = torch.cuda.current_stream(self.device)
                if timers:                                        # the caller came back: its work on the previous batch is enqueued
                    timers[-1][1].record(compute)
                if consumed[k] is not None:                       # ... and device buffer k was used two batches ago
                    side.wait_event(consumed[k])
                tw = time.perf_counter()
                while pending[slot] > 0:
                    try:
                        s, ci, ms, err = done_q.get(timeout=2)
                    except Exception:                             # queue.Empty: is anyone still decoding?
                        dead = [p for p in procs if not p.is_alive()]
                        if dead:
                            raise RuntimeError(f"{len(dead)} decode worker(s) died (exit codes {[p.exitcode for p in dead]}) -- killed by the "
                                               "kernel for memory (-9), out of /dev/shm space behind the upload ring (-7, SIGBUS: "
                                               f"{(_shm_free_bytes() or 0) >> 20} MiB free now), or a crash inside the image library") from None
                        if time.perf_counter() - tw > 600:
                            raise RuntimeError("decode workers made no progress for 600 s") from None
                        continue
                    if err:
                        raise RuntimeError(f"decode worker failed: {err}")
                    metas[s][ci] = ms
                    pending[s] -= 1
                t_wait += time.perf_counter() - tw
                n = min(B, len(self.items) - b * B)
                batch_metas = [m for ci in sorted(metas[slot]) for m in metas[slot][ci]]
                assert len(batch_metas) == n
                n_host += sum(m[3] for m in batch_metas)
                n_dev += n - sum(m[3] for m in batch_metas)
                with torch.cuda.stream(side):
                    for ci in sorted(metas[slot]):                # one copy per chunk: the bytes its images occupy
                        ms = metas[slot][ci]
                        lo = ci * chunk_bytes
                        o, w, h, pt = ms[-1]
                        hi = o + (R * R * 3 if pt else w * h * 3)
                        if pinned:
                            dev_px[k][lo:hi].copy_(host[slot, lo:hi], non_blocking=True)
                        else:
                            staging[:hi - lo].copy_(host[slot, lo:hi])
                            dev_px[k][lo:hi].copy_(staging[:hi - lo], non_blocking=True)
                            side.synchronize()                    # the one staging buffer is reused by the next chunk
                    copied[slot] = torch.cuda.Event()
                    copied[slot].record(side)
                    resizer.run(dev_px[k], batch_metas, dev_u8[k][:n], side)
                    preprocess_u8(dev_u8[k][:n], dev_f16[k][:n], stream=side, mean=self.mean, std=self.std)
                    ready = torch.cuda.Event()
                    ready.record(side)
                compute.wait_event(ready)
                if b + slots < nb:
                    submit(b + slots)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(compute)
                timers.append(ev)
                labels = torch.tensor([l for _, l in self.items[b * B:b * B + n]], dtype=torch.long)
                yield {"img": dev_f16[k][:n], "label": labels}
                consumed[k] = torch.cuda.Event()
                consumed[k].record(torch.cuda.current_stream(self.device))
            if timers:
                timers[-1][1].record(torch.cuda.current_stream(self.device))
            ok = True
        finally:
            torch.cuda.synchronize(self.device)
            for a, z in timers:
                try:
                    gpu_ms += a.elapsed_time(z)
                except (RuntimeError, ValueError):   # an abandoned iteration leaves its last pair open
                    pass
            if ok:
                pool.busy = False                    # every task was answered: the pool serves the next pass
            else:
                _POOLS.pop(pool.key, None)           # tasks may still be in flight: this pool is not reused
                pool.close()
            wall = time.perf_counter() - t0
            self.stats = {"images": len(self.items), "batches": nb, "workers": self.workers, "wall_s": wall, "decode_wait_s": t_wait,
                          "decode_bound_fraction": t_wait / wall if wall > 0 else 0.0,
                          "consumer_gpu_s": gpu_ms / 1e3, "images_per_s": len(self.items) / wall if wall > 0 else 0.0,
                          "encoder_idle_fraction": max(0.0, 1.0 - gpu_ms / 1e3 / wall) if wall > 0 else 0.0, "pinned_ring": bool(pinned),
                          "device_resized": int(n_dev), "host_resized": int(n_host), "raw_cap_bytes": int(pool.cap)}
