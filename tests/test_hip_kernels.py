"""Unit parity of each HIP kernel (through the C ABI's ovmr_debug_* hooks) against a plain torch
fp32/fp64 statement of the same op.  Needs an MI355X: run with `pytest -m gpu`."""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPI_NONE, EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RES, EPI_PATCH, EPI_SCALE = range(6)
GEMM_VARIANTS = [0, 6, 8, 9]    # 0: 128x128 register-staged; 6 / 8: 256-row LDS-DMA tiles (double-buffered / ping-pong K loop; 8 also picks
                                # the 64x64 split-K kernel for latency-bound shapes); 9: the 64x64 split-K kernel wherever it takes the shape
ATTN_VARIANTS = [0, 1, 3, 5]     # 5: 32x32x16 flash kernel for L >= 256 (variant 3 routes ViT-L there); variant 4 lives in the experiment build only


@pytest.fixture(scope="module")
def lib():
    from ovmr_amd import runtime
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return runtime.load_library()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _h(x):  # round through fp16 like the reference's fp16 tensors
    return x.half().float()


def _ref_gemm_f16(A, W, bias, res, pos, epi, scale, rows_in, rows_out):
    acc = A.double() @ W.double().t()
    if epi == EPI_NONE:
        return _h(acc.float())
    if epi == EPI_BIAS:
        return _h((acc + bias.double()).float())
    if epi == EPI_BIAS_QGELU:
        u = _h((acc + bias.double()).float())
        return _h(u * _h(torch.sigmoid(_h(1.702 * u))))
    if epi == EPI_BIAS_RES:
        return _h(_h((acc + bias.double()).float()) + res.float())
    if epi == EPI_SCALE:
        return _h(_h(acc.float()) * scale)
    if epi == EPI_PATCH:
        M, N = acc.shape
        B = M // rows_in
        out = torch.zeros(B * rows_out, N)
        v = _h(_h(acc.float()).reshape(B, rows_in, N) + pos.float()[1:1 + rows_in])
        out.reshape(B, rows_out, N)[:, 1:] = v
        return out
    raise AssertionError


@pytest.mark.parametrize("variant", GEMM_VARIANTS)
@pytest.mark.parametrize("M,N,K,epi", [
    (256, 256, 256, EPI_BIAS), (197 * 3, 768, 768, EPI_BIAS_RES), (1000, 3072, 768, EPI_BIAS_QGELU),
    (130, 2304, 768, EPI_BIAS), (5, 128, 128, EPI_NONE), (64, 1000, 512, EPI_SCALE), (37, 6, 128, EPI_SCALE),
    (4 * 196, 768, 768, EPI_PATCH), (300, 768, 3072, EPI_BIAS_RES), (1, 512, 768, EPI_NONE),
    (2048, 512, 2048, EPI_BIAS_RES), (513, 1536, 512, EPI_BIAS),
])
def test_gemm_f16(lib, variant, M, N, K, epi, with_stats=True):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + epi)
    A = (torch.randn(M, K, generator=g) * 0.5).half()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    bias = (torch.randn(N, generator=g) * 0.1).half()
    rows_in, rows_out = (196, 197) if epi == EPI_PATCH else (0, 0)
    out_rows = M // rows_in * rows_out if epi == EPI_PATCH else M
    res = (torch.randn(out_rows, N, generator=g)).half()
    pos = (torch.randn(197, N, generator=g) * 0.1).half()
    scale = 100.0
    ref = _ref_gemm_f16(A, W, bias, res, pos, epi, scale, rows_in, rows_out)
    d = "cuda"
    Ad, Wd, bd, pd = A.to(d), W.to(d), bias.to(d), pos.to(d)
    C = res.clone().to(d) if epi == EPI_BIAS_RES else torch.zeros(out_rows, N, dtype=torch.float16, device=d)
    # (with EPI_BIAS_RES the hook reads `pos` as the statistics output of the LayerNorm fold: only the tile kernels emit it)
    extra = pd if epi == EPI_PATCH else None
    if epi == EPI_BIAS_RES and with_stats and N % 256 == 0:
        extra = torch.zeros((M, N // 256, 2), dtype=torch.float32, device=d)      # [M][N/256][2] partial (sum, sum of squares) of the stored rows
    rc = lib.ovmr_debug_gemm(0, variant, _p(Ad), _p(Wd), _p(bd), _p(C) if epi == EPI_BIAS_RES else None,
                             _p(extra), _p(C), M, N, K, N, epi, scale, rows_in, rows_out, _s())
    assert rc == 0
    torch.cuda.synchronize()
    got = C.float().cpu()
    if epi == EPI_BIAS_RES and extra is not None and M >= 256:                     # the statistics epilogue of the tile kernels
        st = extra.sum(dim=1).cpu().double()
        np.testing.assert_allclose(st[:, 0].numpy(), got.double().sum(-1).numpy(), rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(st[:, 1].numpy(), (got.double() ** 2).sum(-1).numpy(), rtol=1e-4, atol=1e-2)
    if epi == EPI_PATCH:
        keep = torch.ones(out_rows, dtype=torch.bool)
        keep[::rows_out] = False        # CLS rows are written by a separate kernel
        got, ref = got[keep], ref[keep]
    # fp32 accumulation order differs from fp64: allow one fp16 ulp of the largest magnitude involved
    tol = 2e-3 * max(1.0, float(ref.abs().max()))
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= tol, f"max err {(got - ref).abs().max()}"
    assert float(((got - ref).abs() > tol / 8).float().mean()) < 0.02


@pytest.mark.parametrize("M,N,K,epi", [
    (40, 512, 2048, EPI_BIAS_RES), (3250, 512, 2048, EPI_BIAS_RES), (3250, 512, 512, EPI_BIAS_RES), (1500, 1536, 512, EPI_BIAS),
    (1500, 2048, 512, EPI_BIAS_QGELU), (775, 768, 3072, EPI_BIAS_RES), (775, 3072, 768, EPI_BIAS_QGELU), (125, 512, 512, EPI_NONE),
    (100, 1000, 512, EPI_SCALE), (64, 64, 128, EPI_NONE), (65, 130, 384, EPI_BIAS), (63, 72, 640, EPI_BIAS_RES), (2000, 100, 512, EPI_SCALE),
])
def test_gemm_f16_split_k_small(lib, M, N, K, epi):
    """The 64 x 64 kernel that splits K over its four waves (gemm_f16_small.hip; variant 9 forces it, the default variant 8 picks it
    for these latency-bound shapes): every prefetch depth (K / 128 = 1, 3, 4, 5, 6, 16, 24 steps per wave), ragged M / N edges, the
    in-place residual epilogue, against the fp64 statement."""
    test_gemm_f16(lib, 9, M, N, K, epi, with_stats=False)
    test_gemm_f16(lib, 8, M, N, K, epi, with_stats=False)


def test_gemm_f16_split_k_small_is_deterministic_and_row_independent(lib):
    """The partial tiles are summed in wave order, not by arrival: repeated launches are bit-equal, and a row's values do not depend
    on the rows around it (the same rows as part of a larger M)."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 777, 512, 2048
    A = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    b = (torch.randn(N, generator=g) * 0.1).half().cuda()
    outs = []
    for m in (M, M, 300):
        C = torch.zeros(m, N, dtype=torch.float16, device="cuda")
        assert lib.ovmr_debug_gemm(0, 9, _p(A), _p(W), _p(b), None, None, _p(C), m, N, K, N, EPI_BIAS, 1.0, 0, 0, _s()) == 0
        outs.append(C)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0][:300], outs[2])


@pytest.mark.parametrize("variant", [6, 8])
@pytest.mark.parametrize("M,N,K,epi", [
    (4096, 1024, 768, EPI_BIAS_QGELU), (2500, 2304, 768, EPI_BIAS), (5500, 768, 3072, EPI_BIAS_RES), (4096, 1000, 512, EPI_SCALE),
    (30 * 196, 768, 768, EPI_PATCH), (16400, 256, 128, EPI_NONE), (4097, 1280, 640, EPI_BIAS), (8200, 768, 256, EPI_BIAS_RES),
    (50395, 768, 256, EPI_BIAS_RES), (66000, 1000, 128, EPI_SCALE),      # 2.31 / 4.03 rounds of 256 tiles
    (50432, 768, 768, EPI_BIAS_RES), (50395, 3072, 768, EPI_BIAS_QGELU),       # the batch-256 launch shapes: 2.31 / 9.23 rounds
])
def test_gemm_f16_large_tiles(lib, variant, M, N, K, epi):
    """Shapes with >= 64 tiles of 256 x 256, which take the 256-row tile kernels (variant 6: double-buffered K loop; variant 8:
    ping-pong K loop; K = 640 has an odd number of K-tiles and must fall back), incl. ragged M / N edges."""
    test_gemm_f16(lib, variant, M, N, K, epi)


@pytest.mark.parametrize("M,N,K", [(66000, 1000, 128), (50395, 768, 256), (4100, 1003, 512)])
def test_gemm_fused_row_argmax(lib, M, N, K):
    """EPI_SCALE_ARGMAX (epi 8) against the argmax of the fp16 logits the EPI_SCALE epilogue of the same kernel writes: the
    (maximum, lowest column) pair of every 256-column tile, bit for bit."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.nn.functional.normalize(torch.randn(M, K, generator=g), dim=-1).half().cuda()
    W = torch.nn.functional.normalize(torch.randn(N, K, generator=g), dim=-1).half()
    W[N // 2] = W[3]                                        # an exact tie across tiles: the lower column must win
    W = W.cuda()
    tiles = (N + 255) // 256
    logits = torch.empty((M, N), dtype=torch.float16, device="cuda")
    pairs = torch.zeros((M, tiles, 2), dtype=torch.float32, device="cuda")
    assert lib.ovmr_debug_gemm(0, 8, _p(A), _p(W), None, None, None, _p(logits), M, N, K, N, 5, 10.0, 0, 0, _s()) == 0
    assert lib.ovmr_debug_gemm(0, 8, _p(A), _p(W), None, None, None, _p(pairs), M, N, K, tiles * 2, 8, 10.0, 0, 0, _s()) == 0
    torch.cuda.synchronize()
    lg = torch.nn.functional.pad(logits.float(), (0, tiles * 256 - N), value=float("-inf")).reshape(M, tiles, 256)
    ref_v = lg.amax(dim=-1)
    first = (lg == ref_v[..., None]).float().argmax(dim=-1)          # lowest column holding the maximum
    cols = pairs[..., 1].contiguous().view(torch.int32)
    assert torch.equal(pairs[..., 0], ref_v)
    assert torch.equal(cols.long(), first + 256 * torch.arange(tiles, device="cuda")[None, :])


@pytest.mark.parametrize("M,N,K,epi", [(36, 384, 128, EPI_BIAS), (288, 1536, 512, EPI_BIAS), (288, 512, 2048, EPI_BIAS_RES),
                                       (1000, 2048, 512, EPI_BIAS_QGELU), (7, 128, 512, EPI_BIAS_RES)])
def test_gemm_f32(lib, M, N, K, epi):
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5
    bias, res = torch.randn(N, generator=g) * 0.1, torch.randn(M, N, generator=g)
    acc = (A.double() @ W.double().t() + bias.double())
    ref = {EPI_BIAS: acc, EPI_BIAS_RES: acc + res.double(), EPI_BIAS_QGELU: acc * torch.sigmoid(1.702 * acc)}[epi].float()
    d = "cuda"
    C = res.clone().to(d) if epi == EPI_BIAS_RES else torch.zeros(M, N, device=d)
    Ad, Wd, bd = A.to(d), W.to(d), bias.to(d)
    rc = lib.ovmr_debug_gemm(1, 0, _p(Ad), _p(Wd), _p(bd), _p(C) if epi == EPI_BIAS_RES else None, None, _p(C),
                             M, N, K, N, epi, 1.0, 0, 0, _s())
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(C.cpu().numpy(), ref.numpy(), atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("rows,D,f32", [(1000, 768, 0), (77 * 3, 512, 0), (5, 128, 0), (36, 512, 1), (9, 1024, 0), (3, 2048, 0)])
def test_layernorm(lib, rows, D, f32):
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 3 + 0.5
    gam, bet = torch.randn(D, generator=g) * 0.1 + 1, torch.randn(D, generator=g) * 0.1
    xin = x if f32 else x.half()
    ref = torch.nn.functional.layer_norm(xin.float(), (D,), gam, bet, 1e-5)
    ref = ref if f32 else _h(ref)
    xd = xin.cuda()
    y = torch.empty_like(xd)
    gd, bd = gam.cuda(), bet.cuda()
    assert lib.ovmr_debug_layernorm(f32, _p(xd), _p(y), _p(gd), _p(bd), rows, D, D, _s()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), atol=1e-5 if f32 else 4e-3, rtol=1e-5 if f32 else 2e-3)


def _ref_attention(qkv, B, L, H, causal):
    D = H * 64
    q, k, v = qkv.double().reshape(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.full((L, L), float("-inf"), dtype=torch.float64).triu_(1)
    return (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * L, D).float()


@pytest.mark.parametrize("variant", ATTN_VARIANTS)
@pytest.mark.parametrize("B,L,H,causal", [(3, 197, 12, 0), (2, 5, 2, 0), (4, 77, 8, 1), (5, 9, 2, 1), (1, 577, 16, 0),
                                          (2, 64, 4, 0), (2, 65, 4, 1), (3, 6, 8, 1), (2, 193, 3, 0), (2, 208, 2, 0), (1, 200, 4, 0),
                                          (2, 192, 2, 0), (2, 209, 2, 0),      # 193..208: the single-pass kernel (variant 3)
                                          # >= 256: variant 5 (ViT-L/14: 257, @336px: 577; ragged key blocks and query tiles)
                                          (2, 257, 16, 0), (3, 577, 3, 0), (1, 256, 2, 0), (2, 300, 5, 0), (1, 353, 2, 0), (9, 288, 1, 0)])
def test_attention_f16(lib, variant, B, L, H, causal):
    g = torch.Generator().manual_seed(B * L + H)
    qkv = (torch.randn(B * L, 3 * H * 64, generator=g)).half()
    # spike one key so the online-softmax rescale path is exercised (cdna guide rule 26)
    qkv[L // 2, H * 64:H * 64 + 64] *= 6.0
    if L >= 256:                    # ... and one in a LATE key block of head 0 (the lazily rescaled kernels move their reference there)
        qkv[L - 70, H * 64:H * 64 + 64] *= 9.0
    ref = _ref_attention(qkv.float(), B, L, H, causal)
    qd = qkv.cuda()
    out = torch.zeros(B * L, H * 64, dtype=torch.float16, device="cuda")
    assert lib.ovmr_debug_attention(0, variant, _p(qd), _p(out), B, L, H, causal, _s()) == 0
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) < 6e-3, f"max err {(got - ref).abs().max()}"


@pytest.mark.parametrize("causal", [0, 1])
@pytest.mark.parametrize("B,L,H", [(7, 1, 2), (5, 4, 8), (3, 12, 8), (2, 16, 3), (3, 17, 4), (2, 31, 12), (130, 32, 8), (1, 33, 2)])
def test_attention_f16_short_sequences(lib, B, L, H, causal):
    """Sequences of at most 32 tokens take the one-wave-per-(sequence, head) kernel (attention_short.hip) under every variant but 0:
    against the fp64 statement, and bit for bit against variant 0 (the same arithmetic: one key block, exact row maximum) -- one
    and two query tiles, a ragged last workgroup (pairs not a multiple of four), L = 33 falling back to the general kernel."""
    g = torch.Generator().manual_seed(B * L + H + causal)
    qkv = (torch.randn(B * L, 3 * H * 64, generator=g)).half()
    qkv[L // 2, H * 64:H * 64 + 64] *= 6.0
    ref = _ref_attention(qkv.float(), B, L, H, causal)
    qd = qkv.cuda()
    outs = {}
    for variant in (0, 3):
        out = torch.zeros(B * L, H * 64, dtype=torch.float16, device="cuda")
        assert lib.ovmr_debug_attention(0, variant, _p(qd), _p(out), B, L, H, causal, _s()) == 0
        outs[variant] = out
    torch.cuda.synchronize()
    got = outs[3].float().cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) < 6e-3, f"max err {(got - ref).abs().max()}"
    assert torch.equal(outs[3], outs[0])


@pytest.mark.parametrize("B,L,H", [(6, 18, 8), (3, 6, 2), (2, 66, 8), (1, 10, 12)])
def test_attention_f32(lib, B, L, H):
    g = torch.Generator().manual_seed(B + L + H)
    qkv = torch.randn(B * L, 3 * H * 64, generator=g)
    ref = _ref_attention(qkv, B, L, H, 0)
    qd = qkv.cuda()
    out = torch.zeros(B * L, H * 64, device="cuda")
    assert lib.ovmr_debug_attention(1, 0, _p(qd), _p(out), B, L, H, 0, _s()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("variant", [6, 8, 0, 108])  # 0: the dispatcher must route LN-folding GEMMs to the v5 kernel itself; 108: 8 with the one-rounding QuickGELU (gelu_exact = 0)
@pytest.mark.parametrize("M,D,K1,N2,qgelu,produce", [
    (197 * 3, 768, 768, 2304, 0, True), (1000, 768, 3072, 3072, 1, True), (256, 256, 64, 128, 0, True),
    (513, 512, 2048, 1536, 0, True), (300, 1024, 1024, 4096, 1, True), (462, 512, 0, 2048, 1, False),
    (197 * 28, 768, 768, 3072, 1, True), (5600, 768, 3072, 2304, 0, True),          # >= 64 tiles: the 256-row tile kernels
    (197 * 256 - 19, 768, 256, 768, 1, True),     # the batch-256 launch shapes: 591 tiles = 2.31 rounds in both GEMMs
])
def test_gemm_layernorm_fold(lib, variant, M, D, K1, N2, qgelu, produce):
    """EPI_BIAS_RES with the statistics epilogue, then LayerNorm folded into the next GEMM (common.h EPI_LN_BIAS):
    compared with the reference order of operations -- fp32 LayerNorm rounded to fp16, then nn.Linear (clip/model.py:153-194) --
    and with an unrounded fp64 statement: the folded path must be as close to exact as the reference's own path is."""
    g = torch.Generator().manual_seed(M + D + N2)
    res = torch.randn(M, D, generator=g)
    res[:, 5] *= 20.0                                   # an outlier channel, as CLIP's residual stream has
    res += torch.randn(D, generator=g) * 0.5            # per-channel offsets -> non-zero row means
    res = res.half()
    gamma, beta = 1.0 + 0.3 * torch.randn(D, generator=g), 0.2 * torch.randn(D, generator=g)
    W2 = (torch.randn(N2, D, generator=g) * D ** -0.5).half()
    b2 = (torch.randn(N2, generator=g) * 0.1).half()
    d = "cuda"
    if produce:
        A1 = (torch.randn(M, K1, generator=g) * 0.5).half()
        W1 = (torch.randn(D, K1, generator=g) * K1 ** -0.5).half()
        b1 = (torch.randn(D, generator=g) * 0.1).half()
        x1_ref = _ref_gemm_f16(A1, W1, b1, res, None, EPI_BIAS_RES, 1.0, 0, 0)
        A1d, W1d, b1d = A1.to(d), W1.to(d), b1.to(d)
    else:
        x1_ref = res.float()
    x1 = res.clone().to(d)
    C2 = torch.zeros(M, N2, dtype=torch.float16, device=d)
    gd, bd, W2d, b2d = gamma.to(d), beta.to(d), W2.to(d), b2.to(d)
    rc = lib.ovmr_debug_lnfold(variant, _p(A1d) if produce else None, _p(W1d) if produce else None, _p(b1d) if produce else None,
                               _p(x1), M, D, K1, _p(W2d), _p(gd), _p(bd), _p(b2d), N2, qgelu, _p(x1), _p(C2), _s())
    assert rc == 0
    torch.cuda.synchronize()
    got_x1 = x1.float().cpu()
    tol1 = 2e-3 * max(1.0, float(x1_ref.abs().max()))
    assert float((got_x1 - x1_ref).abs().max()) <= tol1
    # second GEMM: evaluate both statements on the x1 the device produced
    ln64 = torch.nn.functional.layer_norm(got_x1.double(), (D,), gamma.double(), beta.double(), 1e-5)
    ln16 = _h(torch.nn.functional.layer_norm(got_x1, (D,), gamma, beta, 1e-5))

    def tail(u):
        if not qgelu:
            return u
        return u * torch.sigmoid(1.702 * u)
    exact = tail(ln64 @ W2.double().t() + b2.double())
    u_ref = _h((ln16.double() @ W2.double().t() + b2.double()).float())
    ref = _h(u_ref * _h(torch.sigmoid(_h(1.702 * u_ref)))) if qgelu else u_ref
    got = C2.float().cpu()
    assert torch.isfinite(got).all()
    err_got = float((got.double() - exact).pow(2).mean().sqrt())
    err_ref = float((ref.double() - exact).pow(2).mean().sqrt())
    assert err_got <= 1.25 * err_ref + 1e-5, f"folded rms error {err_got:.3e} vs reference path {err_ref:.3e}"
    tol = 4e-3 * max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) <= tol, f"max err {(got - ref).abs().max()}"
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=1)
    assert float((1 - cos).max()) < 1e-5


@pytest.mark.parametrize("variant", [100, 106, 108])      # kernel + 100: the engine's default QuickGELU form (gelu_exact = 0)
@pytest.mark.parametrize("M,N,K", [(1000, 3072, 768), (4096, 1024, 768), (130, 256, 128), (197 * 30, 3072, 768)])
def test_gemm_quickgelu_one_rounding(lib, variant, M, N, K):
    """The one-rounding QuickGELU of the c_fc epilogue, g = h(x * sigmoid(1.702 x)) evaluated in fp32 on the unrounded x = acc + bias
    (common.h quick_gelu_f32x2), against the fp64 function: within ONE fp16 step of the result + 2e-5 (the reference's own fp16 form,
    h(u * h(sigmoid(h(1.702 h(x))))), is up to ~10 steps from the function), and against the reference's fp16 form within the
    absolute bound 8e-3 * max(1, |g|) -- operands with |x| up to ~8."""
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 1.5).half()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    bias = (torch.randn(N, generator=g) * 0.5).half()
    x = A.double() @ W.double().t() + bias.double()
    exact = x * torch.sigmoid(1.702 * x)
    u = _h(x.float())
    ref16 = _h(u * _h(torch.sigmoid(_h(1.702 * u))))
    d = "cuda"
    C = torch.zeros(M, N, dtype=torch.float16, device=d)
    assert lib.ovmr_debug_gemm(0, variant, _p(A.to(d)), _p(W.to(d)), _p(bias.to(d)), None, None, _p(C), M, N, K, N, EPI_BIAS_QGELU,
                               1.0, 0, 0, _s()) == 0
    torch.cuda.synchronize()
    got = C.float().cpu()
    assert torch.isfinite(got).all()
    step = torch.clamp(2.0 ** (torch.floor(torch.log2(exact.abs().clamp_min(2.0 ** -14))) - 10), min=2.0 ** -24)
    # fp32 accumulation against fp64 moves x itself by up to ~1e-5 in absolute terms (K products of ~0.05): an absolute floor beside
    # the step, which is what counts around x = 0 where the steps of g = x / 2 shrink to 2^-24
    assert bool(((got.double() - exact).abs() <= step + 2e-5).all()), float(((got.double() - exact).abs() - step).max())
    assert float(((got - ref16).abs() / ref16.abs().clamp_min(1.0)).max()) <= 8e-3


def test_race_screen_of_hand_synchronised_kernels():
    """GEMM variant 8 (counted vmcnt across barriers, wave rows one barrier apart) and attention variant 3 (LDS-DMA from inline
    asm, hand-placed waits): repeated launches on fixed inputs, L2 / Infinity Cache thrashed in between, must reproduce the first
    launch bit for bit (tools/race_screen.py; the per-shape comparisons against fp64 statements are the tests above)."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "race_screen.py")
    r = subprocess.run([sys.executable, tool, "--reps", "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RACE SCREEN CLEAN" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("B,C,dtype", [(256, 1000, torch.float32), (37, 10, torch.float16), (513, 1001, torch.float32), (64, 21841, torch.float32),
                                       (5, 3, torch.float32), (300, 1000, torch.float16)])
def test_evaluator_counts_kernel_vs_sklearn(lib, tmp_path, B, C, dtype):
    """`Classification` (Dassl.pytorch/dassl/evaluation/evaluator.py:50-138) on the library's own row-argmax-and-count kernel
    (ovmr_eval_counts).  Accuracy, error rate, macro-F1 and BOTH per-class CSVs must equal sklearn's on the host copy of the same outputs --
    with a class that never occurs in the labels, exact ties (lowest column wins, as `mo.max(1)[1]`), rows of identical values, an odd class
    count (rows not 16-byte aligned), several batches accumulated without a host round trip, fp32 probabilities and fp16 logits."""
    from sklearn.metrics import f1_score
    from ovmr_amd.evaluator import Classification
    rng = np.random.default_rng(B * 7 + C)
    absent = C - 1 if C > 3 else None
    gt = rng.integers(0, C - 1 if absent is not None else C, B)
    out = rng.normal(size=(B, C)).astype(np.float32)
    out[np.arange(B), gt] += 2.0 * (rng.random(B) < 0.6)
    out = torch.from_numpy(out).to(dtype)
    # exact ties: the row maximum copied to a LOWER and a HIGHER column; a constant row; a row of -inf
    top = out.argmax(1)
    for r in range(0, B, 5):
        c = int(top[r])
        if c > 0:
            out[r, int(rng.integers(0, c))] = out[r, c]
        if c + 1 < C:
            out[r, int(rng.integers(c + 1, C))] = out[r, c]
    out[1 % B] = 0.25
    out[2 % B] = float("-inf")
    want_pred = out.float().numpy().argmax(1)                                   # numpy: first occurrence of the maximum
    ev = Classification(C, device="cuda")
    dev_out, dev_gt = out.cuda(), torch.from_numpy(gt).cuda()
    cuts = [0, B // 3, B // 3, 2 * B // 3 + 1, B]                               # an empty batch among them
    for a, b in zip(cuts[:-1], cuts[1:]):
        ev.process(dev_out[a:b], dev_gt[a:b])
    tp, n_pred, n_label = (t.numpy() for t in ev.counts())
    assert np.array_equal(n_pred, np.bincount(want_pred, minlength=C)) and np.array_equal(n_label, np.bincount(gt, minlength=C))
    assert np.array_equal(tp, np.bincount(gt[want_pred == gt], minlength=C))
    res = ev.evaluate(str(tmp_path))
    assert res["accuracy"] == pytest.approx(100.0 * float((want_pred == gt).mean()))
    assert res["error_rate"] == pytest.approx(100.0 - res["accuracy"])
    present = np.unique(gt)
    assert res["macro_f1"] == pytest.approx(100.0 * f1_score(gt, want_pred, average="macro", labels=present, zero_division=0))
    if absent is not None:
        assert n_label[absent] == 0 and absent not in present
    f1_rows = open(tmp_path / "f1_per_class.csv").read().strip().split("\n")
    assert f1_rows[0] == "Label,F1" and len(f1_rows) == 1 + len(present)
    per_f1 = 100.0 * f1_score(gt, want_pred, average=None, labels=present, zero_division=0)
    assert [float(x.split(",")[1]) for x in f1_rows[1:]] == pytest.approx(list(per_f1))
    acc_rows = open(tmp_path / "acc_per_class.csv").read().strip().split("\n")
    per_acc = {str(c): 100.0 * float((want_pred[gt == c] == c).mean()) for c in present}
    assert acc_rows[0] == "Label,Acc" and [x.split(",")[0] for x in acc_rows[1:]] == sorted(per_acc)
    assert {x.split(",")[0]: float(x.split(",")[1]) for x in acc_rows[1:]} == pytest.approx(per_acc)
    # the host-side evaluator (device="cpu") holds the same counters
    host = Classification(C, device="cpu")
    host.process(out, torch.from_numpy(gt))
    assert all(np.array_equal(a.numpy(), b) for a, b in zip(host.counts(), (tp, n_pred, n_label)))


def test_evaluator_counts_kernel_edge_cases(lib):
    """ovmr_eval_counts at the ABI: a strided view of a wider matrix (ld > C, unaligned rows), NaN (the largest value, first one wins -- torch's
    rule), int32 labels handed to the Python class, labels outside [0, C) counted apart and raised by the host, argument errors."""
    from ovmr_amd.evaluator import Classification
    C, B = 77, 130
    g = torch.Generator().manual_seed(3)
    wide = torch.randn((B, C + 9), generator=g)
    wide[3, 10] = float("nan")
    wide[3, 40] = float("nan")
    wide[4, 5 + 3] = float("nan")
    view = wide[:, 5:5 + C]                                                     # row stride C + 9, first element 20 bytes into the row
    want = view.argmax(1)                                                       # torch CPU: NaN is the maximum, first occurrence
    assert int(want[3]) == 5 and int(want[4]) == 3
    gt = torch.randint(0, C, (B,), generator=g)
    ev = Classification(C, device="cuda")
    ev.process(wide.cuda()[:, 5:5 + C], gt.int().cuda())
    tp, n_pred, n_label = ev.counts()
    assert torch.equal(n_pred, torch.bincount(want, minlength=C)) and torch.equal(n_label, torch.bincount(gt, minlength=C))
    assert torch.equal(tp, torch.bincount(gt[want == gt], minlength=C))
    bad = gt.clone()
    bad[7], bad[9] = C, -1
    ev.reset()
    ev.process(view.cuda(), bad.cuda())
    with pytest.raises(ValueError, match="2 test label"):
        ev.counts()
    with pytest.raises(ValueError):
        ev.process(torch.zeros(4, C + 1).cuda(), gt[:4].cuda())
    counts = torch.zeros(3 * C + 1, dtype=torch.int32, device="cuda")
    x = torch.zeros(4, C, device="cuda")
    lab = torch.zeros(4, dtype=torch.int64, device="cuda")
    assert lib.ovmr_eval_counts(_p(x), 1, C - 1, _p(lab), 4, C, _p(counts), _s()) == -1      # ld < C
    assert lib.ovmr_eval_counts(_p(x), 2, C, _p(lab), 4, C, _p(counts), _s()) == -1          # dtype
    assert lib.ovmr_eval_counts(None, 1, C, _p(lab), 4, C, _p(counts), _s()) == -1
    assert lib.ovmr_eval_counts(_p(x), 1, C, _p(lab), 0, C, _p(counts), _s()) == 0
    assert lib.ovmr_eval_counts(_p(x), 1, C, _p(lab), 4, C, _p(counts), _s()) == 0
    torch.cuda.synchronize()
    assert int(counts[C]) == 4 and int(counts[0]) == 4 and int(counts[2 * C]) == 4 and int(counts.sum()) == 12   # all-zero rows: column 0


@pytest.mark.parametrize("C,D,n_ctx,n,bound", [(1000, 512, 2, 125, 125), (37, 64, 1, 5, 9), (64, 768, 2, 0, 8), (10, 128, 3, 10, 10)])
def test_pack_and_unpack_rows_of_the_sharded_all_gather(lib, C, D, n_ctx, n, bound):
    """a rank's block of the job's ONE all-gather (ovmr_pack_rows) is bit for bit `ovmr_amd.shard.pack_block` of the
    concatenated rows -- mm | vision | text | visual tokens | the int32 label as two fp16 bit columns, zero padding rows labelled -1 --
    and ovmr_unpack_rows scatters the gathered blocks of all ranks back: every class exactly once (`seen`), stray labels counted apart."""
    from ovmr_amd import synth
    from ovmr_amd.runtime import Engine
    from ovmr_amd.shard import pack_block
    e = Engine(synth.SPECS["tiny"], 2)                                   # (pack / unpack need no weights: only the binding)
    g = torch.Generator(device="cuda").manual_seed(C + D)
    mm, v, t = (torch.randn((C, D), generator=g, device="cuda").half() for _ in range(3))
    tok = torch.randn((C, n_ctx, D), generator=g, device="cuda").half()
    labels = torch.randperm(C, generator=g, device="cuda")[:n]
    block = e.pack_rows(mm, v, t, tok, labels, bound)
    want = pack_block(torch.cat([mm[labels], v[labels], t[labels], tok[labels].flatten(1)], dim=1), labels, bound)
    assert block.shape == want.shape and torch.equal(block.view(torch.int16), want.view(torch.int16))
    with pytest.raises(RuntimeError, match="more than the bound"):
        e.pack_rows(mm, v, t, tok, torch.arange(bound + 1, device="cuda") % C, bound)
    # three "ranks": this block, a block of the remaining classes, an empty block -- and back
    rest = torch.tensor(sorted(set(range(C)) - set(labels.tolist())), device="cuda", dtype=torch.int64)
    blocks = [block, e.pack_rows(mm, v, t, tok, rest, max(1, C - n)), e.pack_rows(mm, v, t, tok, rest[:0], 4)]
    K = (3 + n_ctx) * D
    gathered = torch.cat([b.reshape(-1, K + 2) for b in blocks])
    mm2, v2, t2, tok2, seen = e.unpack_rows(gathered, C, D, n_ctx)
    assert torch.equal(mm2, mm) and torch.equal(v2, v) and torch.equal(t2, t) and torch.equal(tok2, tok)
    assert bool((seen[:C] == 1).all()) and int(seen[C]) == 0
    # a class sent twice, a class never sent, a label outside the vocabulary
    bad = gathered.clone()
    lab = bad[:, K:].contiguous().view(torch.int32).reshape(-1)
    first = int((lab >= 0).nonzero()[0])
    second = int((lab >= 0).nonzero()[1]) if C > 1 else first
    dup, lost = int(lab[first]), int(lab[second])
    lab[second] = dup
    lab[int((lab >= 0).nonzero()[-1])] = C + 5
    bad[:, K:] = lab.view(torch.float16).reshape(-1, 2)
    seen = e.unpack_rows(bad, C, D, n_ctx)[4]
    assert int(seen[dup]) == 2 and int(seen[lost]) == 0 and int(seen[C]) == 1
