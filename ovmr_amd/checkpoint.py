"""Checkpoint ingestion for the hot path (SURVEY.md sections 5.4 and 8f-3).

* CLIP weights: OpenAI's `.pt` files are TorchScript archives; the reference loads them with torch.jit.load and
  falls back to a plain state dict (trainers/mm_classifier_one_prompt.py:29-44, clip/clip.py:117-129).
* Prompt-learner weights: Dassl's save_checkpoint layout `<dir>/prompt_learner/model.pth.tar-<epoch>` =
  {"state_dict", "epoch", "optimizer", "scheduler", "val_result"} plus a `checkpoint` pointer file
  (Dassl.pytorch/dassl/utils/torchtools.py:27-74); MM_CLS_OP.load_model drops token_prefix / token_suffix and
  loads with strict=False (trainers/mm_classifier_one_prompt.py:461-493).
"""
from __future__ import annotations

import os
import os.path as osp
from collections import OrderedDict
from typing import Dict, Optional

import torch


def load_clip_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """State dict of a CLIP checkpoint: TorchScript archive first, plain torch.save second."""
    if not osp.exists(path):
        raise FileNotFoundError(f'CLIP weights not found at "{path}"')
    try:
        sd = torch.jit.load(path, map_location="cpu").eval().state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu")
        if isinstance(sd, dict) and "state_dict" in sd:
            sd = sd["state_dict"]
    return {k: v for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}   # clip/model.py:930-932


def prompt_learner_checkpoint_path(directory: str, epoch: Optional[int] = None, name: str = "prompt_learner") -> str:
    """`<directory>/<name>/model.pth.tar-<epoch>`; without an epoch: "model-best.pth.tar" like the reference (:470-473),
    then the `checkpoint` pointer file written by save_checkpoint."""
    base = osp.join(directory, name)
    if epoch is not None:
        return osp.join(base, f"model.pth.tar-{epoch}")
    best = osp.join(base, "model-best.pth.tar")
    if osp.exists(best):
        return best
    pointer = osp.join(base, "checkpoint")
    if osp.exists(pointer):
        with open(pointer) as f:
            return osp.join(base, f.readline().strip())
    return best


class _Inert:
    """Stands in for every class or function a checkpoint names that is not on the allow list below: constructing it, calling it and
    restoring its state do nothing.  A reference-written checkpoint's scheduler / optimiser objects end up as these."""

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *args, **kwargs):
        return _Inert()

    def __setstate__(self, state):
        pass

    def __reduce__(self):
        return (_Inert, ())


def _restricted_pickle():
    """A pickle-module stand-in for torch.load whose Unpickler resolves ONLY what tensors, containers and scalars need; any other
    global becomes `_Inert`.  Nothing a file names is imported or executed, so a reference-written checkpoint (tensors plus pickled
    scheduler objects) loads without trusting it."""
    import collections
    import pickle
    import types

    allowed_builtins = {"dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex"}
    allowed_torch_utils = {"_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter"}

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module == "collections" and name == "OrderedDict":
                return collections.OrderedDict
            if module == "builtins" and name in allowed_builtins:
                return getattr(__import__("builtins"), name)
            if module == "torch._utils" and name in allowed_torch_utils:
                return getattr(torch._utils, name)
            if module == "torch" and (name.endswith("Storage") or name in ("Size", "device") or isinstance(getattr(torch, name, None), torch.dtype)):
                return getattr(torch, name)
            return _Inert

    def load(f, **kwargs):
        return Unpickler(f, **kwargs).load()

    return types.SimpleNamespace(Unpickler=Unpickler, load=load, __name__="ovmr_restricted_pickle")


def _torch_load(path: str):
    """Tensors, ints and dicts are all the hot path reads from these files, so they are unpickled with weights_only=True: nothing in
    the file can run code.  Checkpoints written by the reference's save_checkpoint also pickle the optimiser state and Dassl's
    scheduler OBJECTS (the warm-up scheduler and its `successor`, Dassl.pytorch/dassl/utils/torchtools.py:27-74), which weights_only
    refuses -- and that is the file `generate_classifier.sh --model-dir ... --load-epoch 30` points at.  Such a file is read with a
    RESTRICTED unpickler (`_restricted_pickle`): tensors and containers load, every other pickled object becomes an inert placeholder,
    no code from the file runs.  OVMR_TRUSTED_CHECKPOINTS=1 in the environment asks for torch's full unpickling instead (what the
    reference's load_checkpoint always does, torchtools.py:66-97) -- only for files whose origin is trusted."""
    import pickle
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        why = str(e).splitlines()[0][:160]
        if os.environ.get("OVMR_TRUSTED_CHECKPOINTS", "0") == "1":
            import warnings
            warnings.warn(f'"{path}" needs full unpickling ({why}); OVMR_TRUSTED_CHECKPOINTS=1: loading it with weights_only=False')
            return torch.load(path, map_location="cpu", weights_only=False)
        try:
            return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_restricted_pickle())
        except Exception as e2:                      # noqa: BLE001
            raise RuntimeError(f'"{path}" holds pickled objects beyond tensors ({why}) and the restricted reader could not load it either '
                               f"({type(e2).__name__}: {e2}).  Set OVMR_TRUSTED_CHECKPOINTS=1 for full unpickling if you trust where "
                               "the file came from, or re-save its state_dict alone") from e2


def load_prompt_learner_checkpoint(directory: str, epoch: Optional[int] = None, name: str = "prompt_learner"):
    """(state dict without token_prefix / token_suffix, epoch recorded in the file, path) of a Dassl checkpoint."""
    path = prompt_learner_checkpoint_path(directory, epoch, name)
    if not osp.exists(path):
        raise FileNotFoundError('Model not found at "{}"'.format(path))          # same message as :477-478
    ckpt = _torch_load(path)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    out = OrderedDict()
    for k, v in sd.items():
        if k.startswith("module."):                                               # torchtools.py:55-60
            k = k[len("module."):]
        if k in ("token_prefix", "token_suffix"):                                 # :482-487
            continue
        out[k] = v
    return out, (ckpt.get("epoch") if isinstance(ckpt, dict) else None), path


def load_prompt_learner_state(directory: str, epoch: Optional[int] = None) -> Dict[str, torch.Tensor]:
    return load_prompt_learner_checkpoint(directory, epoch)[0]


def save_prompt_learner_state(state_dict: Dict[str, torch.Tensor], directory: str, epoch: int,
                              name: str = "prompt_learner") -> str:
    """Writes the Dassl layout (used by tests and by users exporting synthetic weights)."""
    base = osp.join(directory, name)
    os.makedirs(base, exist_ok=True)
    fpath = osp.join(base, f"model.pth.tar-{epoch}")
    torch.save({"state_dict": OrderedDict(state_dict), "epoch": epoch, "optimizer": None, "scheduler": None,
                "val_result": None}, fpath)
    with open(osp.join(base, "checkpoint"), "w") as f:
        f.write(osp.basename(fpath) + "\n")
    return fpath
