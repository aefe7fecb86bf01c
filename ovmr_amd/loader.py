"""Pipelined folder loader: the input side of `ovmr_amd.cli`.

The reference's loaders run 8 DataLoader workers with pinned memory (Dassl.pytorch/dassl/data/data_manager.py:69-113,
configs/trainers/MM_CLS_OP/vit_b16_c4_ep50_imagenet21k_pretrain.yaml:9) and hand fp32 CHW tensors to the model, which moves them
to the GPU and casts to fp16.  Here the split is:

  worker processes (N, spawned, torch-free)   decode + bicubic resize + centre crop -> uint8 HWC, written into a shared,
                                              page-locked ring of batch buffers (150 KB per 224 x 224 image instead of 602 KB fp32)
  side HIP stream                             asynchronous H2D copy of a finished batch, then ovmr_preprocess_u8 (uint8 HWC ->
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


class PipelinedFolderLoader:
    """Iterable of {"img", "label"} dict batches over (path, label) items; see the module docstring.

    With world > 1 the loader is class-sharded like cli.FolderLoader (`presharded = True`).  `stats` after an iteration:
    images, batches, wall_s, decode_wait_s (the caller blocked on the workers), decode_bound_fraction = decode_wait_s / wall_s (the
    share of the job the host spent waiting for JPEG decode: robust whatever streams the consumer uses), consumer_gpu_s (device time
    on the caller's CURRENT stream between a batch becoming available and the caller asking for the next one) and
    encoder_idle_fraction = 1 - consumer_gpu_s / wall_s -- an UPPER bound when the consumer runs the encoder on side streams
    (CustomCLIP.forward_batches: only the staging copy and the hand-over wait fall on the current stream)."""

    def __init__(self, items: Sequence[Tuple[str, int]], batch_size: int, size: int, rank: int = 0, world: int = 1,
                 num_classes: int = 0, workers: int = 8, prefetch: int = 3, device: str = "cuda:0", chunk: int = 8,
                 fast_decode: bool = False, interpolation: str = "bicubic", mean=PIXEL_MEAN, std=PIXEL_STD):
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
        self.stats: Dict[str, float] = {}

    def __len__(self):
        return (len(self.items) + self.bs - 1) // self.bs

    def __iter__(self):
        import multiprocessing as mp
        from multiprocessing import shared_memory
        from . import _decode_worker
        nb = len(self)
        if nb == 0:
            return
        B, R, slots = self.bs, self.size, self.prefetch
        nbytes = slots * B * R * R * 3
        shm = shared_memory.SharedMemory(create=True, size=nbytes)
        host = torch.from_numpy(np.ndarray((slots, B, R, R, 3), dtype=np.uint8, buffer=shm.buf))
        cudart = torch.cuda.cudart()
        pinned = int(cudart.cudaHostRegister(host.data_ptr(), nbytes, 0)) == 0     # DMA straight out of the shared ring
        staging = None if pinned else torch.empty((B, R, R, 3), dtype=torch.uint8).pin_memory()
        ctx = mp.get_context("spawn")               # this process has initialised the GPU: never fork it
        task_q, done_q = ctx.Queue(), ctx.Queue()
        procs = [ctx.Process(target=_decode_worker.worker_main, args=(shm.name, slots, B, R, self.fast, task_q, done_q, self.interpolation), daemon=True)
                 for _ in range(self.workers)]
        for p in procs:
            p.start()
        side = torch.cuda.Stream(self.device)
        dev_u8 = [torch.empty((B, R, R, 3), dtype=torch.uint8, device=self.device) for _ in range(2)]
        dev_f16 = [torch.empty((B, 3, R, R), dtype=torch.float16, device=self.device) for _ in range(2)]
        pending = [0] * slots
        copied = [None] * slots                     # event: the H2D copy out of a ring slot has finished (the slot may be refilled)
        consumed = [None, None]                     # event on the compute stream: the caller is done with device buffer k
        t_wait = gpu_ms = 0.0
        timers: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []

        def submit(b: int) -> None:
            slot = b % slots
            if copied[slot] is not None:
                copied[slot].synchronize()
            chunk_items = self.items[b * B:(b + 1) * B]
            pending[slot] = len(chunk_items)
            for s in range(0, len(chunk_items), self.chunk):
                task_q.put((slot, s, [p for p, _ in chunk_items[s:s + self.chunk]]))

        t0 = time.perf_counter()
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
                        s, n, err = done_q.get(timeout=2)
                    except Exception:                             # queue.Empty: is anyone still decoding?
                        dead = [p for p in procs if not p.is_alive()]
                        if dead:
                            raise RuntimeError(f"{len(dead)} decode worker(s) died (exit codes {[p.exitcode for p in dead]}) -- killed by the "
                                               "kernel for memory, or a crash inside the image library") from None
                        if time.perf_counter() - tw > 600:
                            raise RuntimeError("decode workers made no progress for 600 s") from None
                        continue
                    if err:
                        raise RuntimeError(f"decode worker failed: {err}")
                    pending[s] -= n
                t_wait += time.perf_counter() - tw
                n = min(B, len(self.items) - b * B)
                with torch.cuda.stream(side):
                    if pinned:
                        dev_u8[k][:n].copy_(host[slot, :n], non_blocking=True)
                    else:
                        staging[:n].copy_(host[slot, :n])
                        dev_u8[k][:n].copy_(staging[:n], non_blocking=True)
                    copied[slot] = torch.cuda.Event()
                    copied[slot].record(side)
                    preprocess_u8(dev_u8[k][:n], dev_f16[k][:n], stream=side, mean=self.mean, std=self.std)
                    ready = torch.cuda.Event()
                    ready.record(side)
                if not pinned:
                    copied[slot].synchronize()                    # the one staging buffer is reused by the next batch
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
        finally:
            torch.cuda.synchronize(self.device)
            for a, z in timers:
                try:
                    gpu_ms += a.elapsed_time(z)
                except RuntimeError:                 # an abandoned iteration leaves its last pair open
                    pass
            for _ in procs:
                task_q.put(None)
            for p in procs:
                p.join(timeout=10)
                if p.is_alive():
                    p.terminate()
            if pinned:
                torch.cuda.synchronize(self.device)
                cudart.cudaHostUnregister(host.data_ptr())
            del host
            shm.close()
            shm.unlink()
            wall = time.perf_counter() - t0
            self.stats = {"images": len(self.items), "batches": nb, "workers": self.workers, "wall_s": wall, "decode_wait_s": t_wait,
                          "decode_bound_fraction": t_wait / wall if wall > 0 else 0.0,
                          "consumer_gpu_s": gpu_ms / 1e3, "images_per_s": len(self.items) / wall if wall > 0 else 0.0,
                          "encoder_idle_fraction": max(0.0, 1.0 - gpu_ms / 1e3 / wall) if wall > 0 else 0.0, "pinned_ring": bool(pinned)}
