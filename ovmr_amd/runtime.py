"""ctypes binding to libovmr_hip.so (include/ovmr_hip.h) -- orchestration only.

PyTorch is used for device memory and streams: every tensor handed to the library is a torch
CUDA(=HIP) tensor, the library receives `tensor.data_ptr()` and the current stream.  There is no
CPU fallback: constructing an Engine without the HIP library or without a GPU raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np
import torch

from .synth import ModelSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libovmr_hip.so")

F16, F32, I64, I32 = 0, 1, 2, 3
MODES = {"fusion": 0, "text": 1, "vision": 2, "multimodal": 3}

c_p = ctypes.c_void_p
c_i = ctypes.c_int


class ModelDesc(ctypes.Structure):
    _fields_ = [(n, c_i) for n in ("embed_dim", "image_resolution", "vision_layers", "vision_width",
                                   "vision_patch_size", "context_length", "vocab_size", "transformer_width",
                                   "transformer_layers", "n_ctx", "agg_layers")]


class ResizeJob(ctypes.Structure):
    """ovmr_resize_job (include/ovmr_hip.h)."""
    _fields_ = [("in_offset", ctypes.c_int64), ("tmp_offset", ctypes.c_int64), ("w", c_i), ("h", c_i), ("y0", c_i), ("ny", c_i),
                ("table", c_i), ("ksize_h", c_i), ("ksize_v", c_i), ("passthrough", c_i)]


class TextGroup(ctypes.Structure):
    """ovmr_text_group (include/ovmr_hip.h)."""
    _fields_ = [("prompts_f16", c_p), ("ids", c_p), ("index", c_p), ("n", c_i), ("seq_len", c_i), ("normalize", c_i), ("out_f16", c_p)]


# name -> (restype, argtypes); mirrors include/ovmr_hip.h one to one
SIGNATURES = {
    "ovmr_create": (c_i, [ctypes.POINTER(ModelDesc), ctypes.POINTER(c_p)]),
    "ovmr_destroy": (None, [c_p]),
    "ovmr_last_error": (ctypes.c_char_p, [c_p]),
    "ovmr_version": (ctypes.c_char_p, []),
    "ovmr_set_option": (c_i, [c_p, ctypes.c_char_p, c_i]),
    "ovmr_set_weight": (c_i, [c_p, ctypes.c_char_p, c_p, c_i, c_i, ctypes.POINTER(ctypes.c_int64), c_p]),
    "ovmr_finalize": (c_i, [c_p, c_i, c_i, c_i, c_p]),
    "ovmr_encode_image": (c_i, [c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "ovmr_encode_text_embedded": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "ovmr_encode_text_ids": (c_i, [c_p, c_p, c_i, c_i, c_p, c_i, c_p]),
    "ovmr_encode_text_groups": (c_i, [c_p, ctypes.POINTER(TextGroup), c_i, c_p]),
    "ovmr_embed_tokens": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "ovmr_generate_tokens": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p]),
    "ovmr_assemble_prompts": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "ovmr_xval_counts": (c_i, [c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p]),
    "ovmr_fusion_weights": (c_i, [c_p, c_p, c_p, c_i, ctypes.c_float, c_p, c_p]),
    "ovmr_fused_logits": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    "ovmr_pack_rows": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "ovmr_unpack_rows": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "ovmr_eval_counts": (c_i, [c_p, c_i, ctypes.c_long, c_p, c_i, c_i, c_p, c_p]),
    "ovmr_head_plan": (c_i, [c_p, c_i, c_i]),
    "ovmr_zeroshot_logits": (c_i, [c_p, c_p, c_i, c_p, c_i, c_p, c_p]),
    "ovmr_logit_scale": (ctypes.c_float, [c_p]),
    "ovmr_preprocess_u8": (c_i, [c_p, c_i, c_i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_p, c_p]),
    "ovmr_resize_crop_u8": (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p]),
    "ovmr_encode_chunk": (ctypes.c_int, [c_p]),
    "ovmr_encode_plan": (c_i, [c_p, c_i, ctypes.POINTER(c_i), c_i]),
    "ovmr_flops_per_image": (ctypes.c_double, [c_p]),
    "ovmr_flops_per_image_executed": (ctypes.c_double, [c_p]),
    "ovmr_flops_executed": (ctypes.c_double, [c_p, c_i]),
    "ovmr_flops_per_prompt": (ctypes.c_double, [c_p, c_i]),
    "ovmr_debug_gemm": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, ctypes.c_float, c_i, c_i, c_p]),
    "ovmr_debug_lnfold": (c_i, [c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    "ovmr_debug_layernorm": (c_i, [c_i, c_p, c_p, c_p, c_p, c_i, c_i, ctypes.c_long, c_p]),
    "ovmr_debug_attention": (c_i, [c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
}

_lib = None


def load_library(path: Optional[str] = None) -> ctypes.CDLL:
    """Load libovmr_hip.so and bind every symbol of the C ABI.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("OVMR_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `python -m ovmr_amd.build` (hipcc, gfx950). "
                           "ovmr_amd has no CPU fallback.")
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export the symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class OvmrError(RuntimeError):
    pass


class Engine:
    """One handle = one model on one device, used from one stream at a time."""

    def __init__(self, spec: ModelSpec, n_ctx: int = 2, device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("ovmr_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU fallback")
        self.lib = load_library()
        self.spec = spec
        self.n_ctx = n_ctx
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        desc = ModelDesc(spec.embed_dim, spec.image_resolution, spec.vision_layers, spec.vision_width,
                         spec.vision_patch_size, spec.context_length, spec.vocab_size, spec.transformer_width,
                         spec.transformer_layers, n_ctx, spec.agg_layers)
        h = c_p()
        rc = self.lib.ovmr_create(ctypes.byref(desc), ctypes.byref(h))
        if rc != 0:
            raise OvmrError(f"ovmr_create failed with {rc}")
        self.h = h
        self.finalized = False
        self._options: Dict[str, int] = {}       # what set_option was called with (a twin handle mirrors them: modules.CustomCLIP.forward_batches)
        self._weights_version = 0                # bumped by every set_weight

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.ovmr_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _ck(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.ovmr_last_error(self.h)
            raise OvmrError(f"{what} failed with {rc}: {msg.decode() if msg else ''}")

    def _dev(self, t, dtype=None) -> torch.Tensor:
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        t = t.to(self.device, non_blocking=True)
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()

    def set_option(self, key: str, value: int):
        self._ck(self.lib.ovmr_set_option(self.h, key.encode(), int(value)), "ovmr_set_option")
        self._options[key] = int(value)

    # ------------------------------------------------------------------ weights
    def set_weight(self, name: str, tensor):
        t = self._dev(tensor)
        if t.dtype not in (torch.float16, torch.float32):
            t = t.float()
        shape = (ctypes.c_int64 * max(1, t.dim()))(*t.shape)
        rc = self.lib.ovmr_set_weight(self.h, name.encode(), _ptr(t), F16 if t.dtype == torch.float16 else F32,
                                      t.dim(), shape, _stream())
        self._ck(rc, f"ovmr_set_weight({name})")
        torch.cuda.current_stream().synchronize()   # the source tensor may be freed by the caller right after
        self.finalized = False
        self._weights_version += 1

    def load_state_dict(self, clip_sd: Dict[str, "torch.Tensor"], prompt_learner_sd: Optional[Dict] = None):
        """clip_sd: reference CLIP state dict (clip/model.py:899-936 key names); prompt_learner_sd:
        PromptLearner.state_dict() (trainers/mm_classifier_one_prompt.py:461-493)."""
        skip = ("input_resolution", "context_length", "vocab_size")    # clip/model.py:930-932
        for k, v in clip_sd.items():
            if k not in skip:
                self.set_weight(k, v)
        if prompt_learner_sd is not None:
            for k, v in prompt_learner_sd.items():
                if k in ("token_prefix", "token_suffix"):               # :482-487
                    continue
                self.set_weight("prompt_learner." + k, v)

    def finalize(self, max_images: int = 256, max_prompts: int = 256, max_classes: int = 1024):
        self._ck(self.lib.ovmr_finalize(self.h, max_images, max_prompts, max_classes, _stream()), "ovmr_finalize")
        self.finalized = True

    @property
    def logit_scale(self) -> float:
        return float(self.lib.ovmr_logit_scale(self.h))

    @property
    def encode_chunk(self) -> int:
        """Images per launch sequence of encode_image (chosen by ovmr_finalize so that the GEMM grids are whole rounds of the CUs)."""
        return int(self.lib.ovmr_encode_chunk(self.h))

    def encode_plan(self, B: int) -> list:
        """Image counts of the launch sequences encode_image runs a batch of B images as."""
        buf = (c_i * 4096)()
        n = int(self.lib.ovmr_encode_plan(self.h, int(B), buf, 4096))
        if n < 0:
            raise OvmrError("ovmr_encode_plan: the engine is not finalized")
        return [int(buf[i]) for i in range(min(n, 4096))]

    def flops_per_image(self) -> float:
        return float(self.lib.ovmr_flops_per_image(self.h))

    def flops_per_image_executed(self) -> float:
        return float(self.lib.ovmr_flops_per_image_executed(self.h))

    def flops_executed(self, B: int) -> float:
        """FLOPs ONE encode_image call on B images launches (the plan actually run, the 256-image Q rule per launch sequence)."""
        return float(self.lib.ovmr_flops_executed(self.h, int(B)))

    def flops_per_prompt(self, seq_len: int) -> float:
        return float(self.lib.ovmr_flops_per_prompt(self.h, seq_len))

    # ------------------------------------------------------------------ compute
    def encode_image(self, image: torch.Tensor, normalize: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        image = self._dev(image)
        if image.dtype not in (torch.float16, torch.float32):
            image = image.float()
        B = image.shape[0]
        if out is None:
            out = torch.empty((B, self.spec.embed_dim), dtype=torch.float16, device=self.device)
        self._ck(self.lib.ovmr_encode_image(self.h, _ptr(image), F16 if image.dtype == torch.float16 else F32, B,
                                            _ptr(out), int(normalize), _stream()), "ovmr_encode_image")
        return out

    def encode_text_embedded(self, prompts: torch.Tensor, index: torch.Tensor, seq_len: Optional[int] = None,
                             normalize: int = 0) -> torch.Tensor:
        sl_req = int(seq_len) if seq_len is not None else self.spec.context_length
        if isinstance(index, torch.Tensor) and not index.is_cuda and index.numel() and int(index.max()) >= sl_req:
            raise ValueError(f"seq_len {sl_req} does not reach the largest read-out index {int(index.max())}")
        prompts = self._dev(prompts, torch.float16)
        index = self._dev(index, torch.int32)
        N = prompts.shape[0]
        out = torch.empty((N, self.spec.embed_dim), dtype=torch.float16, device=self.device)
        sl = int(seq_len) if seq_len is not None else self.spec.context_length
        self._ck(self.lib.ovmr_encode_text_embedded(self.h, _ptr(prompts), _ptr(index), N, sl, _ptr(out),
                                                    int(normalize), _stream()), "ovmr_encode_text_embedded")
        return out

    def encode_text_ids(self, ids: torch.Tensor, seq_len: Optional[int] = None, normalize: int = 0) -> torch.Tensor:
        if seq_len is not None and isinstance(ids, torch.Tensor) and not ids.is_cuda and ids.numel():
            eot = int(ids.argmax(dim=-1).max())                 # EOT has the largest id (clip/model.py:831)
            if eot >= int(seq_len):
                raise ValueError(f"seq_len {int(seq_len)} does not reach the EOT token at position {eot}")
        ids = self._dev(ids, torch.int64)
        N = ids.shape[0]
        out = torch.empty((N, self.spec.embed_dim), dtype=torch.float16, device=self.device)
        sl = int(seq_len) if seq_len is not None else self.spec.context_length
        self._ck(self.lib.ovmr_encode_text_ids(self.h, _ptr(ids), N, sl, _ptr(out), int(normalize), _stream()),
                 "ovmr_encode_text_ids")
        return out

    def encode_text_groups(self, groups) -> list:
        """Several prompt families in ONE pass of the text tower (ovmr_encode_text_groups).  Each group is a dict with either
        `prompts` [N, context_length, W] + `index` [N] or `ids` [N, context_length], and `seq_len`, `normalize` as the
        single-group calls.  Returns one [N, embed_dim] fp16 tensor per group."""
        arr = (TextGroup * len(groups))()
        keep, outs = [], []
        for a, g in zip(arr, groups):
            sl = int(g.get("seq_len") or self.spec.context_length)
            # the same host-side checks as the single-family calls (only for tensors that are on the host: no device read-back here)
            src = g.get("ids") if g.get("ids") is not None else g.get("index")
            if isinstance(src, torch.Tensor) and not src.is_cuda and src.numel():
                last = int(src.argmax(dim=-1).max()) if g.get("ids") is not None else int(src.max())
                if last >= sl:
                    raise ValueError(f"seq_len {sl} does not reach the read-out row {last} of a prompt group")
            if g.get("ids") is not None:
                ids = self._dev(g["ids"], torch.int64)
                keep.append(ids)
                a.prompts_f16, a.ids, a.index, n = None, ids.data_ptr(), None, ids.shape[0]
            else:
                pr, ix = self._dev(g["prompts"], torch.float16), self._dev(g["index"], torch.int32)
                keep += [pr, ix]
                a.prompts_f16, a.ids, a.index, n = pr.data_ptr(), None, ix.data_ptr(), pr.shape[0]
            out = torch.empty((n, self.spec.embed_dim), dtype=torch.float16, device=self.device)
            a.n, a.seq_len, a.normalize, a.out_f16 = n, sl, int(g.get("normalize", 0)), out.data_ptr()
            outs.append(out)
        self._ck(self.lib.ovmr_encode_text_groups(self.h, arr, len(groups), _stream()), "ovmr_encode_text_groups")
        return outs

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        ids = self._dev(ids, torch.int64)
        N, L = ids.shape
        out = torch.empty((N, L, self.spec.transformer_width), dtype=torch.float16, device=self.device)
        self._ck(self.lib.ovmr_embed_tokens(self.h, _ptr(ids), N, L, _ptr(out), _stream()), "ovmr_embed_tokens")
        return out

    def generate_tokens(self, feats: torch.Tensor) -> torch.Tensor:
        feats = self._dev(feats, torch.float16)
        Cb, S, D = feats.shape
        out = torch.empty((Cb, self.n_ctx, D), dtype=torch.float32, device=self.device)
        self._ck(self.lib.ovmr_generate_tokens(self.h, _ptr(feats), Cb, S, _ptr(out), _stream()), "ovmr_generate_tokens")
        return out

    def assemble_prompts(self, base: torch.Tensor, labels: Optional[torch.Tensor], tokens: torch.Tensor) -> torch.Tensor:
        base = self._dev(base, torch.float16)
        tokens = self._dev(tokens, torch.float32)
        labels = None if labels is None else self._dev(labels, torch.int64)
        Cb = tokens.shape[0]
        out = torch.empty((Cb, self.spec.context_length, self.spec.transformer_width), dtype=torch.float16, device=self.device)
        self._ck(self.lib.ovmr_assemble_prompts(self.h, _ptr(base), _ptr(labels), _ptr(tokens), Cb, _ptr(out), _stream()),
                 "ovmr_assemble_prompts")
        return out

    def xval_counts(self, feats: torch.Tensor, labels: torch.Tensor, clf: torch.Tensor,
                    tp: torch.Tensor, n_pred: torch.Tensor):
        feats = self._dev(feats, torch.float16)
        labels = self._dev(labels, torch.int32)
        clf = self._dev(clf, torch.float16)
        assert tp.dtype == torch.int32 and n_pred.dtype == torch.int32 and tp.is_contiguous() and n_pred.is_contiguous()
        self._ck(self.lib.ovmr_xval_counts(self.h, _ptr(feats), _ptr(labels), feats.shape[0], _ptr(clf), clf.shape[0],
                                           _ptr(tp), _ptr(n_pred), _stream()), "ovmr_xval_counts")

    def fusion_weights(self, counts: torch.Tensor, n_label: torch.Tensor, tau: float) -> torch.Tensor:
        counts = self._dev(counts, torch.int32)
        n_label = self._dev(n_label, torch.int32)
        C = n_label.shape[0]
        out = torch.empty((C, 3), dtype=torch.float32, device=self.device)
        self._ck(self.lib.ovmr_fusion_weights(self.h, _ptr(counts), _ptr(n_label), C, float(tau), _ptr(out), _stream()),
                 "ovmr_fusion_weights")
        return out

    def pack_rows(self, mm, v, t, tokens, labels, bound: int) -> torch.Tensor:
        """One rank's block of the sharded job's all-gather (ovmr_amd/shard.py pack_block, one launch): rows `labels` of the class-indexed
        fp16 buffers mm / v / t [C, D] and tokens [C, n_ctx, D] + the label bits, [bound, 3 D + n_ctx D + 2] fp16."""
        n, (C, D), n_ctx = int(labels.shape[0]), mm.shape, tokens.shape[1]
        if n > bound:
            raise RuntimeError(f"this rank produced {n} classes, more than the bound {bound} every rank agreed on: the eval-set "
                               "loader yields more than TEST.BATCH_SIZE // NUM_SHOTS classes per batch")
        for x in (mm, v, t, tokens):
            assert x.dtype == torch.float16 and x.is_contiguous() and x.is_cuda
        labels = self._dev(labels, torch.int64)
        block = torch.empty((bound, (3 + n_ctx) * D + 2), dtype=torch.float16, device=self.device)
        self._ck(self.lib.ovmr_pack_rows(_ptr(mm), _ptr(v), _ptr(t), _ptr(tokens), _ptr(labels), n, D, n_ctx, bound, _ptr(block), _stream()),
                 "ovmr_pack_rows")
        return block

    def unpack_rows(self, gathered: torch.Tensor, C: int, D: int, n_ctx: int):
        """The gathered blocks of all ranks -> (mm, v, t [C, D], tokens [C, n_ctx, D], seen int32 [C + 1]); one launch (each result owns its storage: they are saved to files)."""
        gathered = self._dev(gathered, torch.float16)
        assert gathered.shape[1] == (3 + n_ctx) * D + 2
        z = dict(dtype=torch.float16, device=self.device)                                      # (a class no rank sent stays zero; seen says so)
        mm, v, t, tokens = torch.zeros((C, D), **z), torch.zeros((C, D), **z), torch.zeros((C, D), **z), torch.zeros((C, n_ctx, D), **z)
        seen = torch.zeros(C + 1, dtype=torch.int32, device=self.device)
        self._ck(self.lib.ovmr_unpack_rows(_ptr(gathered), gathered.shape[0], C, D, n_ctx, _ptr(mm), _ptr(v), _ptr(t), _ptr(tokens), _ptr(seen),
                                           _stream()), "ovmr_unpack_rows")
        return mm, v, t, tokens, seen

    def fused_logits(self, feats, mm, v, t, w, mode: str = "fusion", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        feats = self._dev(feats, torch.float16)
        cl = [None if x is None else self._dev(x, torch.float16) for x in (mm, v, t)]
        w = None if w is None else self._dev(w, torch.float32)
        C = next(x.shape[0] for x in cl if x is not None)
        if out is None:
            out = torch.empty((feats.shape[0], C), dtype=torch.float32, device=self.device)
        assert out.shape == (feats.shape[0], C) and out.dtype == torch.float32 and out.is_contiguous()
        self._ck(self.lib.ovmr_fused_logits(self.h, _ptr(feats), feats.shape[0], _ptr(cl[0]), _ptr(cl[1]), _ptr(cl[2]),
                                            _ptr(w), C, MODES[mode], _ptr(out), _stream()), "ovmr_fused_logits")
        return out

    def head_plan(self, B: int, C: int) -> int:
        """1: fused_logits runs the one-launch head for B rows x C classes, 0: the GEMM path (their logits may differ by one fp16 step)."""
        return int(self.lib.ovmr_head_plan(self.h, int(B), int(C)))

    def zeroshot_logits(self, feats, text_feats) -> torch.Tensor:
        feats = self._dev(feats, torch.float16)
        text_feats = self._dev(text_feats, torch.float16)
        out = torch.empty((feats.shape[0], text_feats.shape[0]), dtype=torch.float16, device=self.device)
        self._ck(self.lib.ovmr_zeroshot_logits(self.h, _ptr(feats), feats.shape[0], _ptr(text_feats), text_feats.shape[0],
                                               _ptr(out), _stream()), "ovmr_zeroshot_logits")
        return out
