#!/usr/bin/env python3
"""Golden vector for the few-shot exemplar draw: runs the REFERENCE's own
`DatasetBase.generate_fewshot_dataset` (Dassl.pytorch/dassl/data/datasets/base_dataset.py:175-205) after
`set_random_seed(seed)`'s `random.seed(seed)` (Dassl.pytorch/dassl/utils/tools.py, called at train.py:183-186) on a synthetic
list of Datums and records WHICH items it picked, in order.  Runs only in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_fewshot.py      ->  tests/golden/fewshot.npz
"""
import os
import random
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("OVMR_REFERENCE", "/root/reference")


def dataset_items(n_classes=7, seed=3):
    """(path, label) items of a made-up folder dataset: 5..14 images per class, sorted by class then name."""
    r = np.random.default_rng(seed)
    items = []
    for c in range(n_classes):
        for k in range(int(r.integers(5, 15))):
            items.append((f"train/n{c:04d}/img_{k:03d}.JPEG", c))
    return items


def main():
    # base_dataset.py by file path: importing the dassl.data package pulls in torchvision.  Its own module-level imports are gdown
    # (used only by download_data) and dassl.utils.check_isfile (used only by Datum's path check) -- name-only stubs.
    import importlib.util
    sys.modules.setdefault("gdown", types.ModuleType("gdown"))
    dassl, utils = types.ModuleType("dassl"), types.ModuleType("dassl.utils")
    utils.check_isfile = lambda fpath: True
    dassl.utils = utils
    sys.modules.setdefault("dassl", dassl)
    sys.modules.setdefault("dassl.utils", utils)
    spec = importlib.util.spec_from_file_location(
        "ref_base_dataset", os.path.join(REF, "Dassl.pytorch", "dassl", "data", "datasets", "base_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    DatasetBase, Datum = mod.DatasetBase, mod.Datum
    items = dataset_items()
    out = {"paths": np.array([p for p, _ in items]), "labels": np.array([l for _, l in items], dtype=np.int64)}
    for seed in (1, 2, 3):
        for shots in (4, 8):
            data = [Datum(impath=p, label=l, classname=str(l)) for p, l in items]
            random.seed(seed)                                              # what set_random_seed(cfg.SEED) does for this generator
            ds = DatasetBase.__new__(DatasetBase)
            picked = ds.generate_fewshot_dataset(data, num_shots=shots)
            out[f"picked_seed{seed}_shots{shots}"] = np.array([d.impath for d in picked])
    np.savez_compressed(os.path.join(HERE, "fewshot.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
