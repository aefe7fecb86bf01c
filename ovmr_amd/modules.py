"""Host-side mirror of the reference's module API for the OVMR hot path (orchestration only).

Same names, argument meaning, return types and file formats as
trainers/mm_classifier_one_prompt.py (`TextEncoder` :63-91, `PromptLearner` :94-176,
`CustomCLIP` :179-364) and clip/model.py:`build_model` (:899-936); every piece of arithmetic is
a call into libovmr_hip.so through ovmr_amd.runtime.Engine.  Nothing here computes on the CPU and
nothing falls back when the HIP library or the GPU is missing.

Differences a reference user will notice (all documented in DESIGN.md):
  * `clip_model` is a `CLIPModel` (weights + architecture), not an nn.Module;
  * class names are tokenised with the CLIP merge table named by OVMR_BPE_PATH (or ./clip/bpe_simple_vocab_16e6.txt.gz);
    `tokenizer=` (an object with the reference SimpleTokenizer's `encode(str) -> list[int]`) or already tokenised
    prompts (LongTensor [C, 77]) in place of `classnames` are accepted too;
  * the training branch of CustomCLIP.forward (:310-338) is out of scope and raises;
  * with torch.distributed initialised, forward_prompt shards the eval-set batches over ranks,
    all-gathers the classifier rows (RCCL) and all-reduces the F1 counters.
"""
from __future__ import annotations

import os
import os.path as osp
from types import SimpleNamespace
from typing import Dict, Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

from . import synth
from .runtime import Engine
from .synth import ModelSpec, SOT_ID, EOT_ID


# ----------------------------------------------------------------------------- torch.save of host copies, tagged as the device tensors they mirror
_LOCATION_OVERRIDE: Dict[int, int] = {}          # storage data_ptr -> index of the device its values came from


def _override_tagger(storage):
    """torch.serialization location tagger (register_package): storages listed in _LOCATION_OVERRIDE are written with that location --
    "cuda:0" for the page-locked host copy of a device tensor, so that the archive equals a torch.save of the device tensor itself and
    reloads onto the device as the reference's files do (trainers/mm_classifier_one_prompt.py:276-291 saves CUDA tensors)."""
    if not _LOCATION_OVERRIDE:
        return None
    idx = _LOCATION_OVERRIDE.get(storage.data_ptr())
    # a NEW string per storage, as torch's own tagger builds it: the pickler memoises by object identity, and one shared tag object would
    # be written once and referenced afterwards -- other bytes than a torch.save of the device tensors
    return None if idx is None else "cuda:" + str(idx)


torch.serialization.register_package(1, _override_tagger, lambda storage, location: None)


class _saved_as_on_device:
    def __init__(self, tensors, device_index: int):
        self.keys, self.tag = [t.untyped_storage().data_ptr() for t in tensors], int(device_index)

    def __enter__(self):
        for k in self.keys:
            _LOCATION_OVERRIDE[k] = self.tag

    def __exit__(self, *exc):
        for k in self.keys:
            _LOCATION_OVERRIDE.pop(k, None)


# ----------------------------------------------------------------------------- CLIP weights
def infer_spec(state_dict: Dict[str, torch.Tensor], name: str = "from_state_dict") -> ModelSpec:
    """Architecture inference of build_model(), clip/model.py:899-928 (ViT branch only)."""
    if "visual.proj" not in state_dict:
        raise ValueError("only ViT CLIP models are on the hot path (ModifiedResNet is out of scope)")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    patch = state_dict["visual.conv1.weight"].shape[-1]
    grid = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    width = state_dict["ln_final.weight"].shape[0]
    layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    return ModelSpec(name, embed_dim, patch * grid, vision_layers, vision_width, patch, context_length,
                     vocab_size, width, width // 64, layers)


class CLIPModel:
    """Stands where the reference passes `clip_model`: weights in the reference state-dict layout."""

    def __init__(self, state_dict: Dict[str, "torch.Tensor"], spec: Optional[ModelSpec] = None,
                 device: str = "cuda:0"):
        self._sd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in state_dict.items()}
        self.spec = spec or infer_spec(self._sd)
        self.device = torch.device(device)
        self.dtype = torch.float16                      # convert_weights(), clip/model.py:934
        self.visual = SimpleNamespace(output_dim=self.spec.embed_dim, input_resolution=self.spec.image_resolution)
        self.logit_scale = self._sd["logit_scale"].float()
        self.context_length = self.spec.context_length
        self.vocab_size = self.spec.vocab_size
        self._engines: Dict[int, Engine] = {}

    def state_dict(self):
        return self._sd

    def engine(self, n_ctx: int) -> Engine:
        if n_ctx not in self._engines:
            e = Engine(self.spec, n_ctx, str(self.device))
            e.load_state_dict(self._sd)
            self._engines[n_ctx] = e
        return self._engines[n_ctx]

    # CLIP.encode_image / encode_text (clip/model.py:814-833), used by the zero-shot baseline
    def encode_image(self, image, n_ctx: int = 2):
        return _ensure_final(self.engine(n_ctx)).encode_image(image, normalize=False)

    def encode_text(self, text, n_ctx: int = 2):
        return _ensure_final(self.engine(n_ctx)).encode_text_ids(text, normalize=0)


def build_model(state_dict: Dict[str, "torch.Tensor"], device: str = "cuda:0") -> CLIPModel:
    """clip/model.py:899-936."""
    return CLIPModel(state_dict, None, device)


def _ensure_final(e: Engine) -> Engine:
    if not e.finalized:
        if not hasattr(e, "_pl_loaded"):
            raise RuntimeError("prompt-learner weights have not been loaded into the engine yet")
        e.finalize(*getattr(e, "_reserve", (256, 256, 1024)))
    return e


def make_cfg(n_ctx: int = 2, num_shots: int = 16, eval_mode: str = "fusion", eval_tau: float = 10.0,
             output_dir: str = "output_ovmr/generated_classifiers", backbone: str = "ViT-B/16",
             test_batch_size: int = 256, size: int = 224) -> SimpleNamespace:
    """The cfg keys the hot path reads, as a plain namespace tree (no yacs)."""
    return SimpleNamespace(
        TRAINER=SimpleNamespace(COCOOP=SimpleNamespace(N_CTX=n_ctx, PREC="fp16")),
        INPUT=SimpleNamespace(SIZE=(size, size)),
        DATALOADER=SimpleNamespace(TRAIN_X=SimpleNamespace(BATCH_SIZE=1536, N_INS=8), K_TRANSFORMS=1,
                                   TEST=SimpleNamespace(BATCH_SIZE=test_batch_size)),
        DATASET=SimpleNamespace(NUM_SHOTS=num_shots),
        MODEL=SimpleNamespace(BACKBONE=SimpleNamespace(NAME=backbone), INIT_WEIGHTS=""),
        EVAL_MODE=eval_mode, EVAL_TAU=eval_tau, OUTPUT_DIR=output_dir, SEED=1)


_DEFAULT_TOKENIZER = {}


def default_tokenizer():
    """BPETokenizer on the CLIP merge table named by OVMR_BPE_PATH, or found at ./clip/bpe_simple_vocab_16e6.txt.gz (the
    working directory of a reference checkout).  FileNotFoundError when neither exists: the table is not shipped here."""
    from .tokenizer import BPETokenizer
    path = os.environ.get("OVMR_BPE_PATH") or osp.join("clip", "bpe_simple_vocab_16e6.txt.gz")
    if path not in _DEFAULT_TOKENIZER:
        if not osp.exists(path):
            raise FileNotFoundError("class names need the CLIP BPE merge table: set OVMR_BPE_PATH to bpe_simple_vocab_16e6.txt.gz "
                                    "(it is in every CLIP checkout), pass tokenizer=, or pass tokenised prompts [C,77] instead of names")
        _DEFAULT_TOKENIZER[path] = BPETokenizer(path)
    return _DEFAULT_TOKENIZER[path]


def tokenize(texts: Sequence[str], tokenizer, context_length: int = 77) -> torch.Tensor:
    """clip.tokenize (clip/clip.py:187-223) on top of a caller-supplied BPE `tokenizer.encode`."""
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [SOT_ID] + list(tokenizer.encode(t)) + [EOT_ID]
        if len(ids) > context_length:
            raise RuntimeError(f"Input {t} is too long for context length {context_length}")
        out[i, :len(ids)] = torch.tensor(ids)
    return out


# ----------------------------------------------------------------------------- TextEncoder
class TextEncoder:
    """trainers/mm_classifier_one_prompt.py:63-91."""

    def __init__(self, clip_model: CLIPModel, n_ctx: int = 2):
        self.clip_model = clip_model
        self.n_ctx = n_ctx
        self.dtype = torch.float16                     # hard-coded in the reference (:70)

    def forward(self, prompts: torch.Tensor, eos_index: torch.Tensor, seq_len: Optional[int] = None) -> torch.Tensor:
        e = _ensure_final(self.clip_model.engine(self.n_ctx))
        return e.encode_text_embedded(prompts, eos_index, seq_len, normalize=0)

    __call__ = forward


# ----------------------------------------------------------------------------- PromptLearner
class PromptLearner:
    """trainers/mm_classifier_one_prompt.py:94-176 (inference path)."""

    def __init__(self, cfg, classnames, clip_model: CLIPModel, tokenizer=None,
                 state_dict: Optional[Dict[str, "torch.Tensor"]] = None, compute_zero_shot: bool = True,
                 reserve=(256, 256, 1024)):
        self.cfg = cfg
        self.n_ctx = n_ctx = cfg.TRAINER.COCOOP.N_CTX
        self.dtype = torch.float16
        spec = clip_model.spec
        self.clip_model = clip_model
        self.engine = e = clip_model.engine(n_ctx)
        e._reserve = reserve
        dev = e.device
        if isinstance(classnames, torch.Tensor) or isinstance(classnames, np.ndarray):
            tokenized = torch.as_tensor(classnames).long()
            self.name_lens = (tokenized.argmax(-1) - 3).tolist()
        else:
            if tokenizer is None:
                # the reference's module-level `_tokenizer = _Tokenizer()` (:26) reads the merge table that sits next to
                # clip/simple_tokenizer.py; here it comes from OVMR_BPE_PATH or ./clip/bpe_simple_vocab_16e6.txt.gz
                tokenizer = default_tokenizer()
            names = [n.replace("_", " ") for n in classnames]                       # :109
            self.name_lens = [len(tokenizer.encode(n)) for n in names]              # :110
            tokenized = tokenize(["a " + n + "." for n in names], tokenizer, spec.context_length)   # :113,116
        self.num_class = self.n_cls = tokenized.shape[0]
        self._tokenized_host = tokenized
        self._eos_host = tokenized.argmax(-1)
        self.max_eos = int(self._eos_host.max())
        self.tokenized_prompts = tokenized.to(dev)
        self.eos_index = self._eos_host.to(dev)
        vt = torch.from_numpy(synth.template_token_ids(spec.context_length))        # "a ." (:114-115)

        # trainable state: cls_token + aggregator; random CLIP-style init when absent (:145-154)
        if state_dict is None:
            state_dict = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, n_ctx, cfg.SEED).items()}
        self._state: Dict[str, torch.Tensor] = {}
        self.load_state_dict(state_dict, strict=True)
        _ensure_final(e)

        self.prompt_tokens = e.embed_tokens(self.tokenized_prompts)                 # :129,131
        self.visual_prompt_temp = e.embed_tokens(vt.to(dev))                        # :130
        # :118-126.  The reference skips this for num_class >= 5000 and then fails at :265 ("limits the
        # build must lift"); here such vocabularies get their text rows batch by batch inside forward_prompt instead.
        self.zero_shot_classifier = None
        if compute_zero_shot and self.num_class < 5000:
            self.zero_shot_classifier = self.encode_zero_shot(self.tokenized_prompts)

    def encode_zero_shot(self, tokenized: torch.Tensor) -> torch.Tensor:
        """:118-126 -- per class encode_text of its prompt, mean over the single prompt, F.normalize."""
        return self.engine.encode_text_ids(tokenized, seq_len=self.max_eos + 1, normalize=1)

    # nn.Module-like state API
    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._state)

    def load_state_dict(self, state_dict, strict: bool = False):
        expected = set(synth.prompt_learner_keys(self.clip_model.spec))
        sd = {k: v for k, v in state_dict.items() if k not in ("token_prefix", "token_suffix")}     # :482-487
        unknown = set(sd) - expected
        missing = expected - set(sd) - set(self._state)
        if strict and (unknown or missing):
            raise RuntimeError(f"PromptLearner.load_state_dict: missing {sorted(missing)}, unexpected {sorted(unknown)}")
        for k, v in sd.items():
            if k in expected:
                t = (torch.from_numpy(v) if isinstance(v, np.ndarray) else v).detach().float()
                self._state[k] = t
                self.engine.set_weight("prompt_learner." + k, t)
        self.engine._pl_loaded = True
        if not missing:
            self.engine.finalize(*self.engine._reserve)
        return SimpleNamespace(missing_keys=sorted(missing), unexpected_keys=sorted(unknown))

    @property
    def cls_token(self):
        return self._state["cls_token"]

    def eval(self):
        return self

    training = False

    def update_prompts(self, prompt_tokens, ins_tokens):
        """:156-157 (kept for API parity; forward uses the fused assemble kernel)."""
        return torch.cat([prompt_tokens[:, :2], ins_tokens.type(self.prompt_tokens.dtype),
                          prompt_tokens[:, 2:-self.n_ctx]], dim=1)

    def forward(self, exemplar_img_feats: torch.Tensor, label: torch.Tensor, ori_text_len: torch.Tensor):
        """:159-176 -> (mm_prompts_list, mm_lens, v_prompts_list, v_lens, agg_img_token_)."""
        e = self.engine
        mm_lens = ori_text_len + self.n_ctx                                                     # :163
        v_lens = torch.ones_like(ori_text_len, dtype=torch.int32) + self.n_ctx                  # :165
        tokens = e.generate_tokens(exemplar_img_feats)                                          # :167-169
        mm = e.assemble_prompts(self.prompt_tokens, label, tokens)                              # :171
        v = e.assemble_prompts(self.visual_prompt_temp, None, tokens)                           # :173
        return [mm], mm_lens, [v], v_lens, tokens

    __call__ = forward


# ----------------------------------------------------------------------------- two test batches in flight
class _TwoInFlight:
    """The model calls of an evaluation loop, software-pipelined over two handles / two streams (CustomCLIP.forward_batches,
    ZeroshotCLIP.inference_batches).  A user defines `engine`, `device`, `_forward_on(engine, image[, out])` and `_twin_state()`."""

    OVERLAP_MAX_TILES = 1024      # two batches in flight while the narrowest GEMM grid of ONE batch (ceil(rows / 256) x width / 256
                                  # workgroups of 256 x 256) stays within 4 rounds of the 256 CUs; beyond that a batch's partial last round is
                                  # a small share and two batches only contend (profiles/r03u_two_stream_sweep.log: ViT-B/16 128 / 256 / 384
                                  # images +26 / +7 / +4 %, 512 -2 %; ViT-L/14@336px 64 images +8 %, 128 -1 %, 256 -2 %)

    @property
    def OVERLAP_MAX_BATCH(self) -> int:
        """Images per batch up to which forward_batches keeps two batches in flight (settable; default from OVERLAP_MAX_TILES:
        ViT-B/16 443 images, ViT-L/14@336px 113)."""
        if getattr(self, "_overlap_max_batch", None) is not None:
            return self._overlap_max_batch
        sp = self.engine.spec
        tokens = (sp.image_resolution // sp.vision_patch_size) ** 2 + 1
        row_tiles = self.OVERLAP_MAX_TILES // max(1, sp.vision_width // 256)
        return max(1, row_tiles * 256 // tokens)

    @OVERLAP_MAX_BATCH.setter
    def OVERLAP_MAX_BATCH(self, n: int):
        self._overlap_max_batch = int(n)

    IN_FLIGHT = 2                 # test batches in flight in forward_batches / inference_batches: one handle + stream each

    def _twin(self, slot: int = 1) -> Engine:
        """Handle number `slot` (1 ... IN_FLIGHT - 1; 0 is the engine itself): the same weights and options, its own workspace (sized for
        OVERLAP_MAX_BATCH images), used from its own stream.  Rebuilt when the first handle's weights have changed since."""
        e = self.engine
        twins = self.__dict__.setdefault("_twin_engines", {})
        t = twins.get(slot)
        if t is None or t._twin_of_version != e._weights_version:
            t = Engine(e.spec, e.n_ctx, str(e.device))
            t.load_state_dict(*self._twin_state())
            t._pl_loaded = True
            res = getattr(e, "_reserve", (256, 256, 1024))
            t._reserve = (min(res[0], self.OVERLAP_MAX_BATCH), res[1], res[2])
            t._twin_of_version = e._weights_version
            twins[slot] = t
            if slot == 1:
                self._twin_engine = t
        for k, v in e._options.items():
            if t._options.get(k) != v:
                t.set_option(k, v)
        return _ensure_final(t)

    def _run_batches(self, batches, overlap, stable_inputs):
        cur = torch.cuda.current_stream(self.device)

        def hand_over(p):
            out, ev = p
            cur.wait_event(ev)
            out.record_stream(cur)
            return out

        cap = min(self.OVERLAP_MAX_BATCH, getattr(self.engine, "_reserve", (256,))[0])    # what the twin's workspace holds
        try:
            yield from self._forward_batches(batches, overlap, stable_inputs, cap, cur, hand_over)
        finally:                                         # also when the caller abandons the loop: later work on the caller's stream
            for st in getattr(self, "_overlap_streams", ()):     # (another forward on the first handle) is ordered behind what is in flight
                cur.wait_stream(st)

    def _forward_batches(self, batches, overlap, stable_inputs, cap, cur, hand_over):
        import collections
        n = max(2, int(self.IN_FLIGHT))
        pending = collections.deque()                    # (output, event on its stream), oldest first: at most n - 1 while the next is enqueued
        k = 0
        for image in batches:
            image = self.engine._dev(image)
            use = overlap if overlap is not None else image.shape[0] <= cap
            if not use or image.shape[0] > cap:
                while pending:
                    yield hand_over(pending.popleft())
                yield self._forward_on(self.engine, image)
                continue
            if len(getattr(self, "_overlap_streams", ())) < n:
                self._overlap_streams = [torch.cuda.Stream(self.device) for _ in range(n)]
                self._overlap_staging = [None] * n
            slot = k % n
            eng = self.engine if slot == 0 else self._twin(slot)
            st = self._overlap_streams[slot]
            if not stable_inputs:                        # (the previous user of this staging buffer, batch k - n, was handed over already:
                buf = self._overlap_staging[slot]        #  the current stream is ordered behind it)
                if buf is None or buf.shape[1:] != image.shape[1:] or buf.shape[0] < image.shape[0] or buf.dtype != image.dtype:
                    buf = self._overlap_staging[slot] = torch.empty((cap,) + tuple(image.shape[1:]),
                                                                    dtype=image.dtype, device=self.device)
                staged = buf[:image.shape[0]]
                staged.copy_(image)
                image = staged
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                out = self._forward_on(eng, image)
                ev = torch.cuda.Event()
                ev.record(st)
            if stable_inputs:
                image.record_stream(st)
            pending.append((out, ev))
            if len(pending) >= n:
                yield hand_over(pending.popleft())
            k += 1
        while pending:
            yield hand_over(pending.popleft())


# ----------------------------------------------------------------------------- CustomCLIP
class _ImageEncoder:
    """clip_model.visual as CustomCLIP uses it: callable, with .output_dim (:184, :216)."""

    def __init__(self, engine: Engine):
        self.engine = engine
        self.output_dim = engine.spec.embed_dim

    def __call__(self, image):
        return self.engine.encode_image(image, normalize=False)


class CustomCLIP(_TwoInFlight):
    """trainers/mm_classifier_one_prompt.py:179-364 (evaluation / classifier-generation branch)."""

    def __init__(self, cfg, classnames, clip_model: CLIPModel, tokenizer=None,
                 prompt_learner_state: Optional[Dict] = None, reserve=None, distributed: Optional[bool] = None,
                 stream_text: bool = False):
        """distributed: None = shard over the default process group when it has more than one rank; True = take the sharded
        path whenever a process group is initialised (also with ONE rank: the collectives then run through the backend --
        RCCL for "nccl" -- on this rank's own rows, tests/test_hip_distributed.py); False = never.
        stream_text: the zero-shot text rows (:118-126) are not encoded up front by PromptLearner.__init__ but batch by batch inside
        forward_prompt, in the same text-tower pass as the batch's multimodal and vision prompts (what sharded ranks and
        vocabularies of >= 5000 classes always do); same rows, two tower passes fewer per loader batch."""
        self.cfg = cfg
        import torch.distributed as dist
        ready = dist.is_available() and dist.is_initialized()
        if distributed is None:
            distributed = ready and dist.get_world_size() > 1
        if distributed and not ready:
            raise RuntimeError("CustomCLIP(distributed=True) needs an initialised torch.distributed process group")
        self._dist = dist if distributed else None
        if reserve is None:
            reserve = (cfg.DATALOADER.TEST.BATCH_SIZE if hasattr(cfg.DATALOADER, "TEST") else 256, 256, 1024)
        self.prompt_learner = PromptLearner(cfg, classnames, clip_model, tokenizer, prompt_learner_state,
                                            compute_zero_shot=self._dist is None and not stream_text, reserve=reserve)
        self._text_streamed = bool(stream_text) or self._dist is not None
        self.engine = self.prompt_learner.engine
        self.tokenized_prompts = self.prompt_learner.tokenized_prompts
        self.image_encoder = _ImageEncoder(self.engine)
        self.text_encoder = TextEncoder(clip_model, self.prompt_learner.n_ctx)
        self.logit_scale = clip_model.logit_scale
        self.dtype = clip_model.dtype
        self.train_bs = cfg.DATALOADER.TRAIN_X.BATCH_SIZE
        self.num_ins = cfg.DATALOADER.TRAIN_X.N_INS
        self.test_num_ins = cfg.DATASET.NUM_SHOTS
        self.aug_times = cfg.DATALOADER.K_TRANSFORMS
        self.zero_shot_classifier = self.prompt_learner.zero_shot_classifier
        self.mm_classifier = None
        self.visual_classifer = None          # sic -- attribute name of the reference (:225)
        self.fusion_weight = None
        self.device = self.engine.device

    def eval(self):
        return self

    def _twin_state(self):
        return self.prompt_learner.clip_model.state_dict(), self.prompt_learner.state_dict()

    # :200-212
    def get_mm_v_feats(self, mm_prompts, mm_lens, v_prompts, v_lens, text_ids=None):
        """The reference runs the text encoder once on the multimodal prompts and once on the vision prompts (:202-203); here both
        prompt families -- and, with `text_ids` [Cb, 77], the zero-shot text prompts of the same classes (:118-126) -- share ONE pass
        of the tower (Engine.encode_text_groups: each family keeps its own truncated length).  Returns (mm, v) or (mm, v, text)."""
        pl = self.prompt_learner
        n_ctx = pl.n_ctx
        assert len(mm_prompts) == 1 and len(v_prompts) == 1
        # normalize=2: x/x.norm() (:204), mean over the single list element, F.normalize (:210)
        groups = [dict(prompts=mm_prompts[0], index=mm_lens, seq_len=pl.max_eos + n_ctx + 1, normalize=2),
                  dict(prompts=v_prompts[0], index=v_lens, seq_len=2 + n_ctx, normalize=2)]
        if text_ids is not None:
            groups.append(dict(ids=text_ids, seq_len=pl.max_eos + 1, normalize=1))
        return tuple(self.engine.encode_text_groups(groups))

    @staticmethod
    def _batch_images(batch, device):
        image = batch["img"]
        if isinstance(image, (list, tuple)):                                     # K_TRANSFORMS > 1 (:229-234)
            image = torch.cat([im.to(device).unsqueeze(1) for im in image], dim=1).flatten(0, 1)
        return image

    def _reset_generation_state(self):
        """The buffers forward_prompt fills (:216-225): classifier rows, visual tokens, exemplar features, the per-class
        initialised flags, and the text rows when they are produced batch by batch."""
        e, pl, dev = self.engine, self.prompt_learner, self.device
        C, S, D, n_ctx = len(self.tokenized_prompts), self.test_num_ins, e.spec.embed_dim, pl.n_ctx
        f16 = dict(dtype=torch.float16, device=dev)
        self.mm_classifier = torch.zeros((C, D), **f16)
        self.visual_classifer = torch.zeros((C, D), **f16)
        self.inference_text_initialized = torch.zeros(C, dtype=torch.int32, device=dev)
        self.visual_tokens = torch.ones((C, n_ctx, D), **f16)
        self.eval_feat4cls = torch.zeros((C, S, D), **f16)
        # text rows: precomputed by PromptLearner.__init__ (:118-126) for < 5000 classes in one process; otherwise
        # (large vocabularies, or class-sharded ranks) each exemplar batch encodes the prompts of its own classes
        streamed_text = pl.zero_shot_classifier is None or getattr(self, "_text_streamed", False)
        self._text_streamed = streamed_text
        self._text_rows = torch.zeros((C, D), **f16) if streamed_text else self.zero_shot_classifier

    def _generate_local(self, eval_set_loader: Iterable, rank: int = 0, world: int = 1) -> torch.Tensor:
        """Hot loop A (:227-255) over the batches of `eval_set_loader` that belong to `rank` of `world`: exemplar features,
        visual tokens, multimodal / vision (and streamed text) classifier rows written at their class labels into the buffers
        of _reset_generation_state.  Returns the class labels this call produced, in order."""
        e, pl, dev = self.engine, self.prompt_learner, self.device
        S = self.test_num_ins
        streamed_text, text_clf = self._text_streamed, self._text_rows
        local_labels = []
        presharded = bool(getattr(eval_set_loader, "presharded", False))
        cpb = max(1, self.cfg.DATALOADER.TEST.BATCH_SIZE // S)
        for batch_idx, batch in enumerate(eval_set_loader):
            if world > 1 and not presharded and batch["label"].shape[0] // S > cpb:
                # every rank iterates over every batch of a round-robin loader, so every rank raises HERE, before any of them
                # has entered the all-gather whose block size assumes at most `cpb` classes per batch (shard.local_class_bound)
                raise RuntimeError(f"eval-set batch {batch_idx} holds {batch['label'].shape[0] // S} classes, more than "
                                   f"TEST.BATCH_SIZE // NUM_SHOTS = {cpb}")
            if not presharded and batch_idx % world != rank:
                continue
            image = self._batch_images(batch, dev)
            label = batch["label"].to(dev, non_blocking=True)
            num_cls = image.shape[0] // S                                            # :237
            exemplar_label = label.reshape(num_cls, S)[:, 0]                        # :240
            feats = e.encode_image(image, normalize=True).reshape(num_cls, S, -1)   # :243-245
            self.eval_feat4cls[exemplar_label] = feats                              # :247
            mm_p, mm_l, v_p, v_l, tokens = pl(feats, exemplar_label, pl.eos_index[exemplar_label])   # :248
            if streamed_text:                                                       # :249 + the batch's own zero-shot rows (:118-126), one tower pass
                mm, v, t = self.get_mm_v_feats(mm_p, mm_l, v_p, v_l, self.tokenized_prompts[exemplar_label])
                text_clf[exemplar_label] = t
            else:
                mm, v = self.get_mm_v_feats(mm_p, mm_l, v_p, v_l)                   # :249
            self.mm_classifier[exemplar_label] = mm                                 # :251
            self.visual_classifer[exemplar_label] = v                               # :252
            self.inference_text_initialized.index_fill_(0, exemplar_label, 1)       # :254 (`[...] = 1` copies the scalar from the host:
                                                                                    #  a synchronisation in the middle of the head)
            self.visual_tokens[exemplar_label] = tokens.half()                      # :255
            local_labels.append(exemplar_label)
        return torch.cat(local_labels) if local_labels else torch.zeros(0, dtype=torch.long, device=dev)

    @torch.no_grad()
    def forward_prompt(self, eval_set_loader: Iterable, wait_files: bool = True):
        """:214-292.  Returns (mm_classifier, visual_classifer, fusion_weight) and writes
        mm_classifiers.pt / visual_tokens.pt into cfg.OUTPUT_DIR (rank 0 only when distributed).
        wait_files=False (what forward() passes when it generates the classifiers inside a test loop): the files are written behind the
        caller's back (_write_files) and are complete after wait_files(); the default returns with both files on disk, as the reference does."""
        e, pl, dev = self.engine, self.prompt_learner, self.device
        C, S, D, n_ctx = len(self.tokenized_prompts), self.test_num_ins, e.spec.embed_dim, pl.n_ctx
        dist = self._dist
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
        f16 = dict(dtype=torch.float16, device=dev)
        self._reset_generation_state()
        presharded = bool(getattr(eval_set_loader, "presharded", False))
        cpb = max(1, self.cfg.DATALOADER.TEST.BATCH_SIZE // S)
        local = self._generate_local(eval_set_loader, rank, world)
        streamed_text, text_clf = self._text_streamed, self._text_rows

        if dist:
            # ONE all-gather (RCCL) of the packed classifier rows: [mm | v | text | tokens | label bits]; packing and
            # unpacking are one launch each (ovmr_pack_rows / ovmr_unpack_rows) -- a dozen indexing kernels before
            from .shard import all_gather_block, local_class_bound
            bound = local_class_bound(C, world, presharded, cpb)
            block = e.pack_rows(self.mm_classifier, self.visual_classifer, text_clf, self.visual_tokens, local, bound)
            gathered = all_gather_block(block, dist)
            self.mm_classifier, self.visual_classifer, text_clf, self.visual_tokens, seen = e.unpack_rows(gathered, C, D, n_ctx)
            self.inference_text_initialized = (seen[:C] == 1).to(torch.int32)       # every class from exactly one rank
            self._stray_rows = seen[C:]
        if streamed_text:
            self.zero_shot_classifier = self.prompt_learner.zero_shot_classifier = text_clf
        all_initialized = self.inference_text_initialized.bool().all()              # :259 -- read back below, behind the enqueued head:
        if dist:                                                                    # a host round trip here would idle the GPU in front of it
            all_initialized = all_initialized & (self._stray_rows == 0).all()
        self.fusion_weight = self._xval_fusion_weight(local, self.mm_classifier, self.visual_classifer,
                                                      self.zero_shot_classifier, float(self.cfg.EVAL_TAU))   # :261-274
        # rank 0's output files: conversions and device-to-host copies are enqueued HERE, behind the head and in front of the host's one
        # synchronisation (the host is still ahead of the GPU: nothing of it lands on the critical path); the writer starts once the check
        # has passed (the reference asserts before it saves, :259 / :276)
        writer = self._write_files() if rank == 0 and self.cfg.OUTPUT_DIR else None
        assert bool(all_initialized), "a class received no exemplar batch"          # :259
        if writer is not None:
            writer()
            if wait_files:
                self.wait_files()
        return self.mm_classifier, self.visual_classifer, self.fusion_weight

    # ------------------------------------------------------------------ the two output files, off the critical path
    FILE_WRITE_DELAY_S = 0.05     # the writer thread starts its work when wait_files() is called, or this long after the job (see _write_files)
    ASYNC_FILE_WRITE = True       # mm_classifiers.pt / visual_tokens.pt (:276-291) are written by a worker thread behind a side stream while the
                                  # caller goes on (the query loop of a test pass, the next rank's barrier); False: inline, as the reference does.
                                  # The files are the same bytes either way (tests/test_hip_parity.py::test_async_file_write_is_byte_identical).

    def _write_files(self):
        """:276-291.  The saved objects are what the reference saves -- fp32 classifiers and fp16 visual tokens tagged with the device they
        live on -- snapshotted on the caller's stream and copied to page-locked host buffers on a side stream; `torch.save` itself
        (pickling, the zip archive) runs in a worker thread once the copies have landed.  Returns the callable that starts the writer
        (forward_prompt calls it once its completeness check has passed).  wait_files() joins it: forward_prompt calls it before it writes again, MM_CLS_OP.test() / the CLI / bench.py's step
        before they report, and the interpreter joins the (non-daemon) thread at exit."""
        self.wait_files()
        out_dir = self.cfg.OUTPUT_DIR
        mm = {"text_classifier": self.zero_shot_classifier.float(),               # :276-285, all fp32
              "vision_classifier": self.visual_classifer.float(),
              "mm_classifier": self.mm_classifier.float(),
              "fusion_weight": self.fusion_weight.float()}

        def save(mm, vt):
            # each archive is written under its own name inside a scratch directory (torch.save names the archive's records after the
            # file: the bytes are those of a direct torch.save(obj, "<OUTPUT_DIR>/mm_classifiers.pt")) and moved into place whole, so a
            # reader never finds a half-written file under the final name
            os.makedirs(out_dir, exist_ok=True)
            scratch = osp.join(out_dir, f".partial.{os.getpid()}")
            os.makedirs(scratch, exist_ok=True)
            try:
                for obj, name in ((mm, "mm_classifiers.pt"), ({"visual_tokens": vt}, "visual_tokens.pt")):   # :276-291 (visual tokens fp16)
                    torch.save(obj, osp.join(scratch, name))
                    os.replace(osp.join(scratch, name), osp.join(out_dir, name))
            finally:
                import shutil
                shutil.rmtree(scratch, ignore_errors=True)

        if not self.ASYNC_FILE_WRITE:
            vt_now = self.visual_tokens.clone()              # (a copy: `visual_tokens` may be a view of a larger storage, which torch.save would write whole)
            return lambda: save(mm, vt_now)
        import threading
        vt = self.visual_tokens.clone()                      # (a later forward_prompt allocates new buffers, but a caller may write into this one)
        on_gpu = self.device.type == "cuda"                  # (the multi-process CPU tests drive this class with a host-side stand-in engine)
        if on_gpu:
            # The device-to-host copies are ENQUEUED here, by the caller's thread, on a side stream, into page-locked buffers kept from
            # job to job: an asynchronous copy returns at once.  (torch.save on the device tensors from the worker thread copies each
            # storage into pageable memory with a blocking hipMemcpy; while those ran -- 4 ms in all for 8 MB -- the caller's kernel
            # launches queued up behind the runtime, exactly when the GPU had just been drained by the generation's final check: a
            # sharded rank 0 lost 4 ms of its 86 to a write that was meant to cost it nothing.)  The worker thread waits for the copies'
            # event and pickles the HOST tensors; _saved_as_on_device makes torch.save tag their storages with the device the values
            # came from, so the archive is byte for byte the one torch.save of the device tensors writes.
            cur = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            if not hasattr(self, "_file_stream"):
                self._file_stream, self._file_pinned = torch.cuda.Stream(self.device), {}
            side, copied = self._file_stream, torch.cuda.Event()
            host = {}
            with torch.cuda.stream(side):
                side.wait_event(ready)
                for name, t in list(mm.items()) + [("visual_tokens", vt)]:
                    buf = self._file_pinned.get(name)
                    if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
                        buf = self._file_pinned[name] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    buf.copy_(t, non_blocking=True)
                    host[name] = buf
                copied.record(side)
            tag = self.device.index if self.device.index is not None else torch.cuda.current_device()

        go = threading.Event()

        def work():
            try:
                if not on_gpu:
                    return save(mm, vt)
                # The writer yields to the caller's launch loop: it starts its CPU work (waiting on the copies' event, pickling, 8 MB of
                # CRC and tmpfs writes) when the caller asks for the files (wait_files sets `go`) or FILE_WRITE_DELAY_S after the job, whichever
                # comes first.  Started at once, it cost a sharded rank 0 between 0.9 and 2.6 ms of its 86 (two passes on one box,
                # profiles/r06h_*): the first milliseconds after a generation are when the host enqueues the query batches, and a second busy
                # thread in the process slows exactly that; by the time anybody waits for the files the GPU has its work queued.
                go.wait(self.FILE_WRITE_DELAY_S)
                copied.synchronize()                         # (blocks this thread only; mm / vt stay referenced by this closure until then)
                with _saved_as_on_device(list(host.values()), tag):
                    save({k: host[k] for k in mm}, host["visual_tokens"])
            except BaseException as e:                       # noqa: BLE001 -- re-raised by wait_files() on the caller's thread
                self._file_error = e

        def start():
            self._file_error, self._file_go = None, go
            self._file_thread = threading.Thread(target=work, name="ovmr-classifier-files", daemon=False)
            self._file_thread.start()

        return start

    def wait_files(self):
        """Returns once the classifier files of the last forward_prompt are complete on disk (no-op when nothing is in flight); raises what
        the writer raised."""
        t = getattr(self, "_file_thread", None)
        if t is not None:
            self._file_go.set()
            t.join()
            self._file_thread = None
            err, self._file_error = getattr(self, "_file_error", None), None
            if err is not None:
                raise err

    def _xval_fusion_weight(self, local, mm_classifier, v_classifier, t_classifier, tau: float):
        """K18-K20: cross-validation argmax counts on the exemplars this rank encoded (`local` = their class labels,
        features in self.eval_feat4cls), all-reduced over ranks, then F1 and softmax(tau * F1).  Column order mm, v, t."""
        e, dev, dist = self.engine, self.device, self._dist
        C, S = len(self.tokenized_prompts), self.test_num_ins
        counts = torch.zeros((3, 2, C), dtype=torch.int32, device=dev)
        if local.numel():
            rows = self.eval_feat4cls[local].flatten(0, 1)
            row_labels = local.to(torch.int32).unsqueeze(1).expand(-1, S).reshape(-1)   # :261 (repeat_interleave: five launches for the same rows)
            for m, clf in enumerate((mm_classifier, v_classifier, t_classifier)):   # order :272
                e.xval_counts(rows, row_labels, clf, counts[m, 0], counts[m, 1])
        if dist:
            from .shard import all_reduce_counts
            counts = all_reduce_counts(counts, dist)                                # ONE all-reduce
        if getattr(self, "_n_label_key", None) != (C, S):
            self._n_label, self._n_label_key = torch.full((C,), S, dtype=torch.int32, device=dev), (C, S)
        n_label = self._n_label
        self.xval_counts = counts
        return e.fusion_weights(counts, n_label, tau)

    @torch.no_grad()
    def get_fusion_weight(self, eval_set_loader: Iterable, mm_classifier, v_classifier, t_classifier):
        """trainers/coop_mm_classifier.py:235-305: fusion weights for EXTERNALLY supplied classifiers [C, out].
        Encodes the exemplar set, then the same cross-validation F1 preference as forward_prompt with tau fixed at 10
        (:297).  The reference's [S, C, .] permuted feature layout (:277, :285-287) is undone before the row flatten,
        so rows are r = c*S + s with label c exactly as in forward_prompt; nothing is written to disk."""
        e, dev, dist = self.engine, self.device, self._dist
        C, S, D = len(self.tokenized_prompts), self.test_num_ins, e.spec.embed_dim
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
        self.eval_feat4cls = torch.zeros((C, S, D), dtype=torch.float16, device=dev)
        presharded = bool(getattr(eval_set_loader, "presharded", False))
        local_labels = []
        for batch_idx, batch in enumerate(eval_set_loader):
            if not presharded and batch_idx % world != rank:
                continue
            image = self._batch_images(batch, dev)
            label = batch["label"].to(dev, non_blocking=True)
            exemplar_label = label.reshape(image.shape[0] // S, S)[:, 0]              # :261
            self.eval_feat4cls[exemplar_label] = e.encode_image(image, normalize=True).reshape(-1, S, D)   # :263-267
            local_labels.append(exemplar_label)
        local = torch.cat(local_labels) if local_labels else torch.zeros(0, dtype=torch.long, device=dev)
        clfs = [c.to(device=dev, dtype=torch.float16).contiguous() for c in (mm_classifier, v_classifier, t_classifier)]
        self.fusion_weight = self._xval_fusion_weight(local, *clfs, 10.0)
        return self.fusion_weight

    @torch.no_grad()
    def forward(self, image, label=None, eval_set_loader=None, scale_no=None):
        """:294-364, evaluation branch: [B,3,R,R] -> [B,C] fp32."""
        if eval_set_loader is None and self.mm_classifier is None:
            raise NotImplementedError("the training branch of CustomCLIP.forward (autograd) is out of scope; "
                                      "pass eval_set_loader= to generate the classifiers")
        if self.mm_classifier is None:                                              # :341-342 (the features of `image` do not depend on it)
            self.forward_prompt(eval_set_loader, wait_files=False)                  # the files land while this batch and the next ones run
        B = image.shape[0]
        if self.SPLIT_FORWARD and 2 * self.SPLIT_MIN_HALF <= B <= self._split_cap() and self._split_keeps_bits(B):
            return self._forward_split(image)
        return self._forward_on(self.engine, image)

    __call__ = forward

    def _forward_on(self, engine: Engine, image, out=None):
        image_features = engine.encode_image(image, normalize=True)                # :305-307
        mode = self.cfg.EVAL_MODE
        if mode not in ("text", "vision", "multimodal", "fusion"):
            raise ValueError(f"unknown EVAL_MODE {mode}")
        kw = {} if out is None else {"out": out}             # (only the split forward hands an output slice in)
        return engine.fused_logits(image_features, self.mm_classifier, self.visual_classifer,
                                   self.zero_shot_classifier, self.fusion_weight, mode, **kw)

    # ------------------------------------------------------------------ one forward, its two halves in flight
    SPLIT_FORWARD = True          # forward(image) of an UNCHANGED test loop (one model(input) per batch, a host sync per batch): the batch's
    SPLIT_MIN_HALF = 64           # two halves run on two handles / streams, so that the partial last round of GEMM tiles of one half is filled
                                  # by the other's work -- what forward_batches does across batches, without asking the caller for the next batch.
                                  # Same rows bit for bit: a row's arithmetic does not depend on its batch as long as the head takes the
                                  # same implementation (_split_keeps_bits; tests/test_hip_configs.py).

    def _split_keeps_bits(self, B: int) -> bool:
        """The head runs as one launch or as GEMMs + softmax depending on the rows of a call (ovmr_head_plan), and the two may differ by one
        fp16 step in a logit: forward() splits a batch only where both halves take the implementation the whole batch takes, so that
        model(b), forward_batches and SPLIT_FORWARD = False agree bit for bit (e.g. 300 images against 10 000 classes run unsplit)."""
        C, e, half = len(self.tokenized_prompts), self.engine, (B + 1) // 2
        return e.head_plan(B, C) == e.head_plan(half, C) == e.head_plan(B - half, C)

    def _split_cap(self) -> int:
        return min(self.OVERLAP_MAX_BATCH, 2 * getattr(self.engine, "_reserve", (256,))[0])

    def _forward_split(self, image):
        image = self.engine._dev(image)
        B, C = image.shape[0], len(self.tokenized_prompts)
        half = (B + 1) // 2
        cur = torch.cuda.current_stream(self.device)
        if not hasattr(self, "_split_stream"):
            self._split_stream = torch.cuda.Stream(self.device)
        st, twin = self._split_stream, self._twin()
        out = torch.empty((B, C), dtype=torch.float32, device=self.device)
        st.wait_stream(cur)                                  # the image (and `out`) exist for the side stream
        with torch.cuda.stream(st):
            self._forward_on(twin, image[half:], out=out[half:])
        self._forward_on(self.engine, image[:half], out=out[:half])
        cur.wait_stream(st)
        image.record_stream(st)
        out.record_stream(st)
        return out

    @torch.no_grad()
    def forward_batches(self, batches: Iterable, eval_set_loader=None, overlap: Optional[bool] = None, stable_inputs: bool = False):
        """The model calls of the evaluation loop (dassl's test(): one forward per test batch, trainers' model_inference),
        software-pipelined: yields forward(image) for every image batch of `batches`, in order and bit-identical to calling
        forward on each, but with TWO batches in flight -- batch i on one handle and stream, batch i + 1 on a twin handle and a
        second stream -- so that the partial last round of tiles of one batch's launches (batch 256: 197 row tiles x 3 column
        tiles = 2.31 rounds of the 256 CUs on the N = 768 GEMMs) is filled by the other batch's work.  The output of batch i
        is handed over after batch i + 1 has been enqueued; using it on the current stream is ordered after its computation.

        overlap: None = for batches of at most OVERLAP_MAX_BATCH images (measured, ViT-B/16: 256 images 27.7 k -> 29.7 k img/s,
        768 images 30.7 k -> 28.7 k; profiles/r03t_two_stream.log, r03u_two_stream_sweep.log).  stable_inputs: the caller guarantees that a batch's
        tensor is not overwritten before its output has been handed over (a resident data set); otherwise each batch is first
        copied on the current stream, so that loaders which recycle their device buffers (loader.PipelinedFolderLoader)
        stay correct."""
        if self.mm_classifier is None:
            if eval_set_loader is None:
                raise NotImplementedError("pass eval_set_loader= to generate the classifiers")
            self.forward_prompt(eval_set_loader, wait_files=False)
        yield from self._run_batches(batches, overlap, stable_inputs)


class ZeroshotCLIP(_TwoInFlight):
    """trainers/zsclip.py:32-60 (BASELINE config 1): prompts -> text features; raw logits.
    `tokenized_prompts`: LongTensor [C, 77] -- clip.tokenize of CUSTOM_TEMPLATES[dataset].format(classname) (:42-45; tokenize() above with a
    BPE tokenizer produces them from names)."""

    def __init__(self, clip_model: CLIPModel, tokenized_prompts: torch.Tensor, n_ctx: int = 2, reserve=(256, 256, 1024)):
        self.clip_model = clip_model
        self.engine = e = clip_model.engine(n_ctx)
        self.device = e.device
        if not hasattr(e, "_pl_loaded"):                   # (the engine's handle also serves CustomCLIP: it wants the aggregator's weights)
            self._pl_state = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(clip_model.spec, n_ctx, 0).items()}
            e.load_state_dict({}, self._pl_state)
            e._pl_loaded = True
        if not e.finalized:
            e._reserve = tuple(reserve)
            e.finalize(*reserve)
        self.tokenized_prompts = tokenized_prompts
        self.text_features = e.encode_text_ids(tokenized_prompts, normalize=1)      # :47-50

    def _twin_state(self):
        pl = getattr(self, "_pl_state", None)
        if pl is None:
            pl = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(self.clip_model.spec, self.engine.n_ctx, 0).items()}
        return self.clip_model.state_dict(), pl

    def _forward_on(self, engine: Engine, image, out=None):
        f = engine.encode_image(image, normalize=True)                              # :56-57
        return engine.zeroshot_logits(f, self.text_features)                        # :58-59

    def model_inference(self, image):
        return self._forward_on(self.engine, image)

    @torch.no_grad()
    def inference_batches(self, batches: Iterable, overlap: Optional[bool] = None, stable_inputs: bool = False):
        """model_inference for every image batch of the test loop (Dassl.pytorch/dassl/engine/trainer.py:461-482), in order and bit-identical
        to calling it per batch, two batches in flight on two handles / streams (as CustomCLIP.forward_batches)."""
        yield from self._run_batches(batches, overlap, stable_inputs)
