"""`MM_CLS_OP`: the evaluation / classifier-generation side of the reference's trainer class, on the HIP path.

Mirrors the names, call order and error behaviour of `MM_CLS_OP` (trainers/mm_classifier_one_prompt.py:367-493) and of
the `SimpleTrainer` methods it inherits for testing (Dassl.pytorch/dassl/engine/trainer.py:460-521), so a script written
against the reference

    trainer = build_trainer(cfg)                  # TRAINER_REGISTRY -> MM_CLS_OP(cfg)
    trainer.load_model(args.model_dir, epoch=args.load_epoch)
    trainer.test()

runs unchanged with `trainer = ovmr_amd.trainer.MM_CLS_OP(cfg, dm, clip_weights=...)`.  Everything that needs autograd
(forward_backward, the optimiser, save_model) is out of scope and raises.

`dm` is anything with the four attributes the reference's DataManager (Dassl.pytorch/dassl/data/data_manager.py:116-170)
hands the trainer: `dataset.classnames`, `test_loader`, `val_loader` (may be None) and `eval_set_loader`
(`RandomClassSampler` batches: S consecutive rows per class).
"""
from __future__ import annotations

import collections
import os
import os.path as osp
import random
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import checkpoint, modules
from .evaluator import Classification

TRAINER_REGISTRY: Dict[str, type] = {}


class MM_CLS_OP:
    def __init__(self, cfg, dm, clip_weights=None, tokenizer=None, device: str = "cuda:0"):
        self.check_cfg(cfg)
        self.cfg, self.dm, self.device = cfg, dm, torch.device(device)
        self._clip_weights, self._tokenizer = clip_weights, tokenizer
        self._models: "OrderedDict[str, object]" = OrderedDict()
        self.test_loader = dm.test_loader
        self.val_loader = getattr(dm, "val_loader", None)
        self.eval_set_loader = getattr(dm, "eval_set_loader", None)      # data_manager.py:157-170
        self.num_classes = len(dm.dataset.classnames)
        self.epoch = 0
        self.output_dir = cfg.OUTPUT_DIR
        self.build_model()
        self.evaluator = Classification(self.num_classes, list(dm.dataset.classnames), device=str(self.device))

    # :369-370
    def check_cfg(self, cfg):
        assert cfg.TRAINER.COCOOP.PREC in ["fp16", "fp32", "amp"]

    # :372-419 (inference-relevant part: CLIP weights -> CustomCLIP -> optional INIT_WEIGHTS -> register "prompt_learner")
    def build_model(self):
        cfg = self.cfg
        random.seed(cfg.SEED)
        classnames = self.dm.dataset.classnames
        print(f"Loading CLIP (backbone: {cfg.MODEL.BACKBONE.NAME})")
        w = self._clip_weights
        if w is None:
            raise FileNotFoundError("MM_CLS_OP needs clip_weights= (an OpenAI CLIP .pt file or a state dict): "
                                    "there is no download on this path (clip/clip.py:29-70 fetches it in the reference)")
        sd = checkpoint.load_clip_state_dict(w) if isinstance(w, (str, os.PathLike)) else w
        clip_model = modules.build_model(sd, device=str(self.device))
        if cfg.TRAINER.COCOOP.PREC != "fp16":
            # the reference calls clip_model.float() here (:380-382) and then fails inside forward_prompt, whose buffers are
            # fp16 (:216-225): only fp16 evaluation exists
            raise RuntimeError("the OVMR evaluation path is fp16-only (trainers/mm_classifier_one_prompt.py:216-225)")
        print("Building custom CLIP")
        self.model = modules.CustomCLIP(cfg, classnames, clip_model, tokenizer=self._tokenizer)
        init = getattr(cfg.MODEL, "INIT_WEIGHTS", "")
        if init:                                                              # load_pretrained_weights (:403-404)
            ckpt = checkpoint._torch_load(init)
            self.model.prompt_learner.load_state_dict(ckpt["state_dict"] if "state_dict" in ckpt else ckpt, strict=False)
        self.register_model("prompt_learner", self.model.prompt_learner)     # :410

    def register_model(self, name, model, optim=None, sched=None):
        self._models[name] = model

    def get_model_names(self, names=None):
        return list(self._models.keys()) if names is None else list(names)

    # :454-459 / trainer.py:510-518
    def parse_batch_train(self, batch):
        return batch["img"].to(self.device), batch["label"].to(self.device)

    def parse_batch_test(self, batch):
        return batch["img"].to(self.device), batch["label"].to(self.device)

    # :461-493 -- the file handling lives in ovmr_amd.checkpoint (Dassl layout, "module." prefixes, dropped token buffers)
    def load_model(self, directory, epoch=None):
        if not directory:
            print("Note that load_model() is skipped as no pretrained model is given")
            return
        for name in self.get_model_names():
            state, saved_epoch, path = checkpoint.load_prompt_learner_checkpoint(directory, epoch, name)
            print(f'Loading weights to {name} from "{path}" (epoch = {saved_epoch})')
            self._models[name].load_state_dict(state, strict=False)

    # trainer.py:504-508
    def model_inference(self, input, scale_no=0, label=None):
        if self.eval_set_loader is not None:
            return self.model(input, eval_set_loader=self.eval_set_loader, scale_no=scale_no, label=label)
        return self.model(input, label=label)

    # trainer.py:460-482 (DATASET.REGION_AUG False)
    @torch.no_grad()
    def test(self, split=None):
        self.evaluator.reset()
        if split is None:
            split = getattr(getattr(self.cfg, "TEST", None), "SPLIT", "test")
        if split == "val" and self.val_loader is not None:
            data_loader = self.val_loader
        else:
            split = "test"
            data_loader = self.test_loader
        print(f"Evaluate on the *{split}* set")
        labels = collections.deque()                 # model_inference's forwards (:504-508), two batches in flight (CustomCLIP.forward_batches)

        def inputs():
            for batch in data_loader:
                input, label = self.parse_batch_test(batch)
                labels.append(label)
                yield input

        for output in self.model.forward_batches(inputs(), eval_set_loader=self.eval_set_loader):
            self.evaluator.process(output, labels.popleft())
        self.model.wait_files()                      # mm_classifiers.pt / visual_tokens.pt were written while the test set ran
        results = self.evaluator.evaluate(self.output_dir or None)
        return list(results.values())[0]

    def forward_backward(self, batch):
        raise NotImplementedError("training (autograd through CustomCLIP.forward, :310-338) is out of scope of the HIP hot path")

    train = save_model = forward_backward


TRAINER_REGISTRY["MM_CLS_OP"] = MM_CLS_OP


def build_trainer(cfg, dm, **kw):
    """dassl.engine.build_trainer: look the class up by cfg.TRAINER.NAME."""
    name = getattr(cfg.TRAINER, "NAME", "MM_CLS_OP")
    if name not in TRAINER_REGISTRY:
        raise ValueError(f"unknown trainer {name!r}; on the hot path: {sorted(TRAINER_REGISTRY)}")
    return TRAINER_REGISTRY[name](cfg, dm, **kw)
