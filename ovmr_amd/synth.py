"""Seeded, platform-independent synthetic weights and inputs for the OVMR hot path.

There are no CLIP weights, no aggregator checkpoint and no datasets offline
(SURVEY.md headline fact 5), so every parity test and the benchmark run on
tensors produced here.  The generator is counter based (splitmix64 -> Box-Muller
in float64, rounded once to float32), so the build container, the GPU box and the
golden-vector script all see identical tensors without shipping 172 MB of weights.

Weight names follow the reference state dict (clip/model.py:899-936 for CLIP,
trainers/mm_classifier_one_prompt.py:138-154 for the prompt learner) and the init
scales follow clip/model.py:773-800 and trainers/mm_classifier_one_prompt.py:145-154.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, Optional

import numpy as np

_U64 = np.uint64
_MASK = (1 << 64) - 1

SOT_ID = 49406  # clip/simple_tokenizer.py: "<|startoftext|>"
EOT_ID = 49407  # "<|endoftext|>" is the largest id, so ids.argmax() finds it (clip/model.py:831)
TOK_A = 320     # "a</w>"   (probed)
TOK_DOT = 269   # ".</w>"
CONTEXT_LENGTH = 77


@dataclass(frozen=True)
class ModelSpec:
    """Architecture hyper-parameters, named as in clip/model.py:717-731."""
    name: str
    embed_dim: int
    image_resolution: int
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_heads: int
    transformer_layers: int
    agg_layers: int = 4          # hard-coded in the reference, trainers/mm_classifier_one_prompt.py:140

    @property
    def vision_heads(self) -> int:  # clip/model.py:745
        return self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size

    @property
    def vision_tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def agg_heads(self) -> int:  # trainers/mm_classifier_one_prompt.py:141
        return self.embed_dim // 64

    def asdict(self):
        return asdict(self)


SPECS: Dict[str, ModelSpec] = {
    # build_model() rules, clip/model.py:903-928
    "ViT-B/16": ModelSpec("ViT-B/16", 512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "ViT-B/32": ModelSpec("ViT-B/32", 512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    "ViT-L/14": ModelSpec("ViT-L/14", 768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12),
    "ViT-L/14@336px": ModelSpec("ViT-L/14@336px", 768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
    # parity-test sizes (all intermediates cheap on CPU)
    "tiny": ModelSpec("tiny", 128, 32, 2, 128, 16, 77, 49408, 128, 2, 2),
    "small": ModelSpec("small", 256, 64, 3, 256, 16, 77, 49408, 256, 4, 3),
    # checkpoint-format fixtures (tests/golden/gen_checkpoints.py): small enough to commit as a TorchScript archive
    "micro": ModelSpec("micro", 64, 32, 1, 64, 16, 77, 512, 64, 1, 1),
    # the ViT-L/14 HEAD (embed_dim = text width = 768, 12 text heads, 768-wide aggregator) behind a 2-layer vision tower
    "head768": ModelSpec("head768", 768, 64, 2, 256, 16, 77, 49408, 768, 12, 2),
}


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _key(name: str, seed: int) -> int:
    return (fnv1a64(name) ^ ((seed * 0xD1342543DE82EF95) & _MASK)) & _MASK


def uniform01(name: str, n: int, seed: int, offset: int = 0) -> np.ndarray:
    """float64 in (0,1), element i depends only on (name, seed, offset+i)."""
    k = _U64(_key(name, seed))
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _splitmix64(_splitmix64(idx + k) ^ k)
    return ((z >> _U64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(name: str, shape, seed: int, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """N(mean, std) float32 tensor; chunked so 100M-element tensors stay small in RAM."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    out = np.empty(n, dtype=np.float32)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        u1 = uniform01(name + "#1", m, seed, s)
        u2 = uniform01(name + "#2", m, seed, s)
        r = np.sqrt(-2.0 * np.log(u1))
        out[s:s + m] = (mean + std * r * np.cos(2.0 * np.pi * u2)).astype(np.float32)
    return out.reshape(tuple(shape))


def randint(name: str, n: int, lo: int, hi: int, seed: int) -> np.ndarray:
    """int64 in [lo, hi)."""
    u = uniform01(name, n, seed)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64).clip(lo, hi - 1)


def _block(sd, prefix, width, layers, seed, jitter, attn_std, proj_std, fc_std):
    for i in range(layers):
        p = f"{prefix}{i}."
        sd[p + "attn.in_proj_weight"] = normal(p + "attn.in_proj_weight", (3 * width, width), seed, attn_std)
        sd[p + "attn.out_proj.weight"] = normal(p + "attn.out_proj.weight", (width, width), seed, proj_std)
        sd[p + "mlp.c_fc.weight"] = normal(p + "mlp.c_fc.weight", (4 * width, width), seed, fc_std)
        sd[p + "mlp.c_proj.weight"] = normal(p + "mlp.c_proj.weight", (width, 4 * width), seed, proj_std)
        for nm, n in (("attn.in_proj_bias", 3 * width), ("attn.out_proj.bias", width),
                      ("mlp.c_fc.bias", 4 * width), ("mlp.c_proj.bias", width),
                      ("ln_1.bias", width), ("ln_2.bias", width)):
            sd[p + nm] = (normal(p + nm, (n,), seed, 0.05) if jitter else np.zeros(n, np.float32))
        for nm in ("ln_1.weight", "ln_2.weight"):
            sd[p + nm] = (normal(p + nm, (width,), seed, 0.1, 1.0) if jitter else np.ones(width, np.float32))


def clip_state_dict(spec: ModelSpec, seed: int = 0, jitter: bool = False,
                    logit_scale: float = float(np.log(100.0))) -> Dict[str, np.ndarray]:
    """fp32 CLIP state dict with the reference's key names.

    jitter=False: the CLIP init rules (LN gamma 1, beta 0, biases 0) used by bench.py.
    jitter=True : biases / LN affine parameters are randomised too so that parity
                  tests exercise every parameter.
    logit_scale defaults to ln(100), the trained CLIP value.
    """
    sd: Dict[str, np.ndarray] = {}
    W, P, L = spec.vision_width, spec.vision_patch_size, spec.vision_tokens
    vs = W ** -0.5  # clip/model.py:369
    sd["visual.class_embedding"] = normal("visual.class_embedding", (W,), seed, vs)
    sd["visual.positional_embedding"] = normal("visual.positional_embedding", (L, W), seed, vs)
    sd["visual.proj"] = normal("visual.proj", (W, spec.embed_dim), seed, vs)
    sd["visual.conv1.weight"] = normal("visual.conv1.weight", (W, 3, P, P), seed, (3 * P * P) ** -0.5)
    for nm in ("visual.ln_pre", "visual.ln_post"):
        sd[nm + ".weight"] = normal(nm + ".weight", (W,), seed, 0.1, 1.0) if jitter else np.ones(W, np.float32)
        sd[nm + ".bias"] = normal(nm + ".bias", (W,), seed, 0.05) if jitter else np.zeros(W, np.float32)
    _block(sd, "visual.transformer.resblocks.", W, spec.vision_layers, seed, jitter,
           W ** -0.5, (W ** -0.5) * ((2 * spec.vision_layers) ** -0.5), (2 * W) ** -0.5)

    T = spec.transformer_width
    sd["token_embedding.weight"] = normal("token_embedding.weight", (spec.vocab_size, T), seed, 0.02)
    sd["positional_embedding"] = normal("positional_embedding", (spec.context_length, T), seed, 0.01)
    sd["ln_final.weight"] = normal("ln_final.weight", (T,), seed, 0.1, 1.0) if jitter else np.ones(T, np.float32)
    sd["ln_final.bias"] = normal("ln_final.bias", (T,), seed, 0.05) if jitter else np.zeros(T, np.float32)
    sd["text_projection"] = normal("text_projection", (T, spec.embed_dim), seed, T ** -0.5)
    sd["logit_scale"] = np.array(logit_scale, dtype=np.float32)
    _block(sd, "transformer.resblocks.", T, spec.transformer_layers, seed, jitter,
           T ** -0.5, (T ** -0.5) * ((2 * spec.transformer_layers) ** -0.5), (2 * T) ** -0.5)
    return sd


def prompt_learner_keys(spec: ModelSpec):
    """Keys of PromptLearner.state_dict(): cls_token + 12 tensors per aggregator block."""
    keys = ["cls_token"]
    for i in range(spec.agg_layers):
        for s in ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                  "ln_1.weight", "ln_1.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight",
                  "mlp.c_proj.bias", "ln_2.weight", "ln_2.bias"):
            keys.append(f"aggregator.resblocks.{i}.{s}")
    return keys


def prompt_learner_state_dict(spec: ModelSpec, n_ctx: int = 2, seed: int = 0,
                              jitter: bool = False) -> Dict[str, np.ndarray]:
    """The 1 + 12*agg_layers trainable tensors of PromptLearner.state_dict()."""
    sd: Dict[str, np.ndarray] = {}
    D = spec.embed_dim
    c = normal("cls_token", (n_ctx, D), seed)
    sd["cls_token"] = (c / np.linalg.norm(c, axis=-1, keepdims=True)).astype(np.float32)
    _block(sd, "aggregator.resblocks.", D, spec.agg_layers, seed + 7919, jitter,
           D ** -0.5, (D ** -0.5) * ((2 * spec.agg_layers) ** -0.5), (2 * D) ** -0.5)
    return sd


def class_token_ids(num_classes: int, seed: int = 4321, context_length: int = CONTEXT_LENGTH) -> np.ndarray:
    """Tokenised '"a " + name + "."' prompts with random class-name ids.

    Row c = [SOT, 'a', id_1..id_l, '.', EOT, 0...], l ~ U{1..4}; EOS index = l + 3.
    """
    out = np.zeros((num_classes, context_length), dtype=np.int64)
    lens = randint("name_len", num_classes, 1, 5, seed)
    ids = randint("name_ids", num_classes * 4, 0, SOT_ID, seed).reshape(num_classes, 4)
    for c in range(num_classes):
        l = int(lens[c])
        row = [SOT_ID, TOK_A] + [int(t) for t in ids[c, :l]] + [TOK_DOT, EOT_ID]
        out[c, :len(row)] = row
    return out


def template_token_ids(context_length: int = CONTEXT_LENGTH) -> np.ndarray:
    """Tokenised visual template "a ." (trainers/mm_classifier_one_prompt.py:114)."""
    out = np.zeros((1, context_length), dtype=np.int64)
    out[0, :4] = [SOT_ID, TOK_A, TOK_DOT, EOT_ID]
    return out


def images(n: int, resolution: int, seed: int = 1234, class_ids: Optional[np.ndarray] = None,
           class_strength: float = 0.0, start: int = 0, tile: int = 0) -> np.ndarray:
    """[n,3,R,R] float32 ~ N(0,1) (post-normalisation statistics).

    With class_ids and class_strength > 0 each image is
    sqrt(1-s^2)*noise + s*pattern[class], so exemplars of one class are correlated (used by parity tests only).
    tile = 0: the class pattern is i.i.d. per pixel (a random-weight ViT averages it away: its features barely
    depend on the class).  tile = t > 0: the class pattern is ONE [3,t,t] patch repeated over the image, which
    survives the near-uniform attention of a random-weight ViT, so features of different classes separate.
    Image i depends only on (seed, start+i): shards can generate their own slice.
    """
    px = 3 * resolution * resolution
    out = np.empty((n, px), dtype=np.float32)
    for i in range(n):
        out[i] = normal(f"img{start + i}", (px,), seed)
    if class_ids is not None and class_strength > 0.0:
        s = float(class_strength)
        for i in range(n):
            if tile > 0:
                reps = -(-resolution // tile)
                pat = normal(f"tile{int(class_ids[i])}", (3, tile, tile), seed + 1)
                pat = np.tile(pat, (1, reps, reps))[:, :resolution, :resolution].reshape(px)
            else:
                pat = normal(f"pattern{int(class_ids[i])}", (px,), seed + 1)
            out[i] = np.sqrt(1.0 - s * s) * out[i] + s * pat
    return out.reshape(n, 3, resolution, resolution)


def trained_like_statistics(sd: Dict[str, np.ndarray], spec: ModelSpec, seed: int = 17) -> np.ndarray:
    """In place: give a CLIP-init state dict the statistics a TRAINED CLIP ViT shows and init weights do not (every other parity fixture is
    CLIP-init): a few residual-stream channels with massive activations (tens of standard deviations, appearing after an early MLP and
    carried to the end), LayerNorm gains that are tiny on those channels and spread over 0.1 .. 4 elsewhere, peaky attention (large q / k),
    c_fc pre-activations far into both QuickGELU tails.  Returns the hot channels.  Same draws on every platform (numpy PCG64 stream of
    `seed`): tests/golden/gen_golden.py feeds these weights to the real reference (`hot.npz`), the GPU test regenerates them."""
    rng = np.random.default_rng(seed)
    W = spec.vision_width
    hot = rng.choice(W, size=3, replace=False)
    sd["visual.transformer.resblocks.0.mlp.c_proj.bias"][hot] = np.array([42.0, -31.0, 18.0], dtype=np.float32)
    for i in range(spec.vision_layers):
        q = f"visual.transformer.resblocks.{i}."
        for ln in ("ln_1", "ln_2"):
            g = np.exp(rng.normal(0.0, 0.8, W)).clip(0.1, 4.0).astype(np.float32)
            g[hot] = 0.03
            sd[q + ln + ".weight"] = g
            sd[q + ln + ".bias"] = rng.normal(0.0, 0.3, W).astype(np.float32)
        sd[q + "attn.in_proj_weight"][:2 * W] *= 2.5                 # q and k rows: logits ~6x wider
        sd[q + "mlp.c_fc.bias"] = rng.normal(0.0, 2.5, 4 * W).astype(np.float32)
    sd["visual.ln_post.weight"][hot] = 0.05
    return hot


def align_state_dicts(sd: Dict[str, np.ndarray], pl: Dict[str, np.ndarray], spec: ModelSpec, gain: float) -> None:
    """Give random-init weights the ONE property of trained OVMR weights the cross-validation step (K18-K20) relies on:
    classifier rows that point towards their own class's image features.  In place, on fp32 state dicts:
    an identity component `gain * I` is added to the value projection and to out_proj of every text-tower and aggregator
    block and to text_projection, so attention copies (normalised) token content forward: visual tokens ~ mean exemplar
    feature, the vision / multimodal classifier rows ~ the visual tokens.  Everything else stays random, so the
    fixtures still exercise every parameter.  Needs transformer_width == embed_dim (true for every CLIP ViT)."""
    T, E = spec.transformer_width, spec.embed_dim
    assert T == E, "align_state_dicts needs transformer_width == embed_dim"
    eye = np.eye(T, dtype=np.float32) * np.float32(gain)
    for i in range(spec.transformer_layers):
        p = f"transformer.resblocks.{i}."
        sd[p + "attn.in_proj_weight"][2 * T:] += eye
        sd[p + "attn.out_proj.weight"] += eye
    sd["text_projection"] += eye
    for i in range(spec.agg_layers):
        p = f"aggregator.resblocks.{i}."
        pl[p + "attn.in_proj_weight"][2 * E:] += eye
        pl[p + "attn.out_proj.weight"] += eye
