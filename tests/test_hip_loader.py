"""Input side of the runner on the GPU: ovmr_resize_crop_u8 against PIL (bit for bit), ovmr_preprocess_u8 against
the reference's test transform, and the pipelined loader (decode-only worker processes -> pinned shared ring -> side-stream upload +
resize / crop + preprocess) against the plain loader."""
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
    # a second pass over the same loader object works (the worker pool and the shared ring are kept between passes)
    assert sum(b["img"].shape[0] for b in pipe) == 21
    loader.close_pools()


def _sizes(n, rng):
    """Input sizes for the resize tests: the common photograph shapes, squares, exact-size and one-edge-equal inputs (PIL skips a
    pass there), up- and down-scaling, extreme aspect ratios, tiny images -- then random ones."""
    fixed = [(500, 375), (375, 500), (224, 224), (224, 300), (300, 224), (336, 336), (100, 37), (37, 100), (1200, 224), (224, 1200),
             (1, 1), (2, 900), (900, 2), (223, 225), (225, 223), (640, 480), (1024, 768), (64, 64), (449, 448), (2000, 1500)]
    out = list(fixed)
    while len(out) < n:
        out.append((int(rng.integers(3, 1400)), int(rng.integers(3, 1400))))
    return out[:n]


@pytest.mark.parametrize("R,interpolation,n", [(224, "bicubic", 210), (336, "bicubic", 40), (224, "bilinear", 60), (64, "bicubic", 40)])
def test_resize_crop_u8_bit_equal_to_pil(R, interpolation, n):
    """ovmr_resize_crop_u8 (PIL's two integer passes on the GPU over host-built tables, ovmr_amd/resize.py) against PIL itself --
    `_decode_worker.load_u8`: Image.resize of the smaller edge to R + centre crop, what torchvision's Resize / CenterCrop do in
    _build_transform_test (Dassl.pytorch/dassl/data/transforms/transforms.py:495-526) -- on n input sizes, EVERY byte equal.
    The images mix noise (every weight matters), flat saturated areas (clip8 at 0 / 255) and smooth ramps."""
    from PIL import Image
    from ovmr_amd import _decode_worker, loader
    rng = np.random.default_rng(R + len(interpolation) + n)
    frames = []
    for i, (w, h) in enumerate(_sizes(n, rng)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 3 == 1:
            a[: h // 2, : w // 2] = 255
            a[h // 2:, w // 2:] = 0
        elif i % 3 == 2:
            a = ((np.arange(w)[None, :, None] * 3 + np.arange(h)[:, None, None] * 5 + np.arange(3)[None, None, :] * 40) % 256).astype(np.uint8)
        frames.append(a)
    want = np.stack([_decode_worker.load_u8(Image.fromarray(a), R, False, interpolation) for a in frames])
    for lo in range(0, n, 64):                                   # several launches, several sizes per launch
        got = loader.resize_crop_u8(frames[lo:lo + 64], R, interpolation).cpu().numpy()
        torch.cuda.synchronize()
        for k in range(got.shape[0]):
            h, w = frames[lo + k].shape[:2]
            assert np.array_equal(got[k], want[lo + k]), f"{w} x {h} -> {R} ({interpolation}): {int((got[k] != want[lo + k]).sum())} bytes differ"


def test_pipelined_loader_resizes_on_the_device_and_keeps_its_pool(tmp_path):
    """Decode-only workers: raw frames up to the ring's share per image are resized and cropped on the GPU, larger ones (and every image
    under `nearest`) by PIL in the worker -- both kinds in one batch, bit-equal to the plain loader; the worker pool and the ring are
    reused by the next pass and by a second loader with the same geometry (one start-up), and closed by close_pools()."""
    from ovmr_amd import cli, loader
    rng = np.random.default_rng(11)
    items = _write_images(tmp_path, 3, 7, rng)                         # sizes 40..400: frames of 5 KB .. 480 KB
    plain = list(cli.FolderLoader(items, 8, 64))
    cap = 150 * 1024                                                   # frames above 150 KB take the host path
    pipe = loader.PipelinedFolderLoader(items, 8, 64, workers=2, prefetch=2, chunk=3, raw_cap_bytes=cap)
    for b, a in zip(pipe, plain):
        assert torch.equal(a["img"].half(), b["img"].cpu()) and torch.equal(a["label"], b["label"])
    st = pipe.stats
    assert st["device_resized"] > 0 and st["host_resized"] > 0 and st["device_resized"] + st["host_resized"] == 21
    pools = dict(loader._POOLS)
    assert len(pools) == 1
    pids = [p.pid for p in next(iter(pools.values())).procs]
    for b, a in zip(pipe, plain):                                      # second pass: the same pool
        assert torch.equal(a["img"].half(), b["img"].cpu())
    other = loader.PipelinedFolderLoader(items[:13], 5, 64, workers=2, prefetch=2, chunk=3, raw_cap_bytes=cap)   # smaller batches fit the same ring
    assert sum(b["img"].shape[0] for b in other) == 13
    assert [p.pid for p in next(iter(loader._POOLS.values())).procs] == pids and len(loader._POOLS) == 1
    # nearest: PIL's nearest is another algorithm -- every image is resized in the worker
    near = loader.PipelinedFolderLoader(items, 8, 64, workers=2, prefetch=2, chunk=3, interpolation="nearest")
    plain_near = list(cli.FolderLoader(items, 8, 64, interpolation="nearest"))
    for b, a in zip(near, plain_near):
        assert torch.equal(a["img"].half(), b["img"].cpu())
    assert near.stats["device_resized"] == 0
    # an abandoned pass does not poison the next one
    it = iter(loader.PipelinedFolderLoader(items, 8, 64, workers=2, prefetch=2, chunk=3, raw_cap_bytes=cap))
    next(it)
    it.close()
    again = loader.PipelinedFolderLoader(items, 8, 64, workers=2, prefetch=2, chunk=3, raw_cap_bytes=cap)
    for b, a in zip(again, plain):
        assert torch.equal(a["img"].half(), b["img"].cpu())
    loader.close_pools()
    assert not loader._POOLS


def test_resize_crop_u8_edges():
    """n = 0 is a no-op; NULL pointers / non-positive sizes come back as OVMR_E_ARG (no launch); an already R x R frame passes through
    PIL's identity (both passes skipped there, unit weights here) unchanged; pass-through jobs are copied."""
    import ctypes
    from ovmr_amd import loader, runtime
    lib = runtime.load_library()
    z = ctypes.c_void_p(0)
    assert lib.ovmr_resize_crop_u8(z, z, 0, z, z, 0, z, 224, z) == 0
    buf = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    p = ctypes.c_void_p(buf.data_ptr())
    assert lib.ovmr_resize_crop_u8(z, p, 1, p, p, 1, p, 224, z) != 0           # no pixels
    assert lib.ovmr_resize_crop_u8(p, p, 1, p, p, 1, p, 0, z) != 0             # R = 0
    assert lib.ovmr_resize_crop_u8(p, p, 1, p, z, 5, p, 224, z) != 0           # an intermediate is needed but not given
    rng = np.random.default_rng(3)
    same = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    got = loader.resize_crop_u8([same], 64).cpu().numpy()[0]
    assert np.array_equal(got, same)
    # pass-through: the arena already holds the crop
    r = loader.DeviceResizer(64, "bicubic")
    arena = torch.from_numpy(same.reshape(-1).copy()).cuda()
    out = torch.empty((1, 64, 64, 3), dtype=torch.uint8, device="cuda")
    r.run(arena, [(0, 64, 64, 1)], out, torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy()[0], same)
