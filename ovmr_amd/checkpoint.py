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


def _torch_load(path: str):
    """Tensors, ints and dicts are all the hot path reads from these files, so they are unpickled with weights_only=True: nothing in
    the file can run code.  Checkpoints written by the reference's save_checkpoint also pickle the optimiser state and Dassl's
    scheduler OBJECTS (the warm-up scheduler and its `successor`), which weights_only refuses.  Such a file is loaded only when the
    caller says it is trusted -- OVMR_TRUSTED_CHECKPOINTS=1 in the environment -- because full unpickling executes whatever the file
    asks for (what the reference's load_checkpoint always does, Dassl.pytorch/dassl/utils/torchtools.py:66-97); otherwise this raises
    with the way out."""
    import pickle
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        why = str(e).splitlines()[0][:160]
        if os.environ.get("OVMR_TRUSTED_CHECKPOINTS", "0") != "1":
            raise RuntimeError(f'"{path}" holds pickled objects beyond tensors ({why}).  Loading it means FULL unpickling, which can '
                               "execute code from the file: set OVMR_TRUSTED_CHECKPOINTS=1 if you trust where it came from, or "
                               "re-save its state_dict alone (torch.save({'state_dict': ckpt['state_dict'], 'epoch': ...}))") from e
        import warnings
        warnings.warn(f'"{path}" needs full unpickling ({why}); OVMR_TRUSTED_CHECKPOINTS=1: loading it with weights_only=False')
        return torch.load(path, map_location="cpu", weights_only=False)


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
