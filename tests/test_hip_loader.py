"""Input side of the runner on the GPU: ovmr_preprocess_u8 against the reference's test transform, and the
pipelined loader (worker processes -> pinned shared ring -> side-stream upload + preprocess) against the plain loader."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_images(root, n_classes, per_class, rng):
    from PIL import Image
    items = []
    for c in range(n_classes):
        d = root / f"n{c:02d}"
        d.mkdir(parents=True)
        for i in range(per_class):
            h, w = int(rng.integers(40, 400)), int(rng.integers(40, 400))
            arr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            arr[: h // 2] = (arr[: h // 2].astype(np.int32) // 4 + 40 * c).clip(0, 255).astype(np.uint8)
            path = d / (f"{i}.png" if i % 3 else f"{i}.jpg")
            Image.fromarray(arr).save(path, **({"quality": 90} if path.suffix == ".jpg" else {}))
            items.append((str(path), c))
    return items


@pytest.mark.parametrize("R", [224, 64, 336])
def test_preprocess_u8_bit_equal_to_the_test_transform(tmp_path, R):
    """uint8 HWC (decode + bicubic resize + centre crop, done on the host) -> ovmr_preprocess_u8 must equal the fp32 tensor of
    cli.test_transform (= _build_transform_test: ToTensor, Normalize) rounded to fp16 -- bit for bit."""
    from PIL import Image
    from ovmr_amd import _decode_worker, cli, loader
    rng = np.random.default_rng(R)
    items = _write_images(tmp_path, 2, 5, rng)
    u8 = np.stack([_decode_worker.load_u8(p, R) for p, _ in items])
    ref = torch.stack([cli.test_transform(Image.open(p), R) for p, _ in items]).half()
    got = loader.preprocess_u8(torch.from_numpy(u8).cuda())
    torch.cuda.synchronize()
    assert got.dtype == torch.float16 and tuple(got.shape) == (len(items), 3, R, R)
    assert torch.equal(got.cpu(), ref)
    # every byte value in every channel
    ramp = torch.arange(256, dtype=torch.uint8).repeat(3 * R * R // 256 + 1)[: R * R * 3].reshape(1, R, R, 3).contiguous()
    x = ramp[0].permute(2, 0, 1).float().div(255.0)
    want = ((x - torch.tensor(cli.PIXEL_MEAN).view(3, 1, 1)) / torch.tensor(cli.PIXEL_STD).view(3, 1, 1)).half()
    assert torch.equal(loader.preprocess_u8(ramp.cuda())[0].cpu(), want)


def test_pipelined_loader_equals_the_plain_loader(tmp_path):
    """Same batches, bit for bit, as cli.FolderLoader (PIL + torch on the host, fp32 -> fp16 cast as the model does at
    trainers/mm_classifier_one_prompt.py:243), ragged last batch included; the statistics the runner prints are filled in."""
    from ovmr_amd import cli, loader
    rng = np.random.default_rng(7)
    items = _write_images(tmp_path, 3, 7, rng)                         # 21 images, batches of 8: 8 + 8 + 5
    plain = list(cli.FolderLoader(items, 8, 64))
    pipe = loader.PipelinedFolderLoader(items, 8, 64, workers=3, prefetch=2, chunk=3)
    n = 0
    for b, a in zip(pipe, plain):                                      # (pipe first: zip then runs it to its end)
        assert b["img"].is_cuda and b["img"].dtype == torch.float16
        assert torch.equal(a["img"].half(), b["img"].cpu()) and torch.equal(a["label"], b["label"])
        n += 1
    assert n == len(plain) == len(pipe) == 3
    st = pipe.stats
    assert st["images"] == 21 and st["batches"] == 3 and st["wall_s"] > 0 and 0.0 <= st["encoder_idle_fraction"] <= 1.0
    # a second pass over the same loader object works (workers and the shared ring are per iteration)
    assert sum(b["img"].shape[0] for b in pipe) == 21
