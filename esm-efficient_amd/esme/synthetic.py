"""Deterministic synthetic checkpoints and packed-batch generators.

There is no network on the build or GPU boxes, so released weights are
unavailable; benchmarks, parity tests and golden fixtures all use weights drawn
from `numpy.random.Generator(PCG64(...))` (never torch RNG, so every box
regenerates bit-identical tensors) and written in the reference's checkpoint
layout (writer in the reference: workflow/common/safetensor.py:66-79).

Batch generators follow SURVEY.md §8(d): uniform-S batches and the
proteome-like greedy token-budget packing rule of esme/data.py:32-54.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Tuple

import numpy as np
import torch

# name -> (kind, num_layers, embed_dim, attention_heads)
MODEL_ZOO = {
    'esm2_8m':   ('esm2', 6, 320, 20),
    'esm2_35m':  ('esm2', 12, 480, 20),
    'esm2_150m': ('esm2', 30, 640, 20),
    'esm2_650m': ('esm2', 33, 1280, 20),
    'esm2_3b':   ('esm2', 36, 2560, 40),
    'esm2_15b':  ('esm2', 48, 5120, 40),
    'esm1b':     ('esm1b', 33, 1280, 20),
    'esm1v':     ('esm1v', 33, 1280, 20),
    'esmc_300m': ('esmc', 30, 960, 15),
    'esmc_600m': ('esmc', 36, 1152, 18),
}


def swiglu_width(embed_dim: int, expand: float = 8 / 3) -> int:
    """ESM-C FFN width: expand*E rounded up to a multiple of 256 (attention.py:218-219)."""
    return int(((expand * embed_dim) + 255) // 256 * 256)


def tensor_shapes(kind: str, num_layers: int, embed_dim: int) -> Dict[str, Tuple[int, ...]]:
    """Checkpoint tensor names and shapes, in file order, for an ESM-2 or ESM-C model."""
    E = embed_dim
    shapes: Dict[str, Tuple[int, ...]] = {}
    if kind in ('esm2', 'esm1b', 'esm1v'):
        V, F = 33, 4 * E
        shapes['embed_tokens.weight'] = (V, E)
        if kind != 'esm2':                      # learned positions (+ LayerNorm before, ESM-1b)
            shapes['embed_positions.weight'] = (4098, E)
            if kind == 'esm1b':
                shapes['emb_layer_norm_before.weight'] = (E,)
                shapes['emb_layer_norm_before.bias'] = (E,)
        for i in range(num_layers):
            p = f'layers.{i}.'
            shapes[p + 'self_attn.norm.weight'] = (E,)
            shapes[p + 'self_attn.norm.bias'] = (E,)
            for n in ('q', 'k', 'v', 'out'):
                shapes[p + f'self_attn.{n}.weight'] = (E, E)
                shapes[p + f'self_attn.{n}.bias'] = (E,)
            shapes[p + 'final.0.weight'] = (E,)
            shapes[p + 'final.0.bias'] = (E,)
            shapes[p + 'final.1.weight'] = (F, E)
            shapes[p + 'final.1.bias'] = (F,)
            shapes[p + 'final.3.weight'] = (E, F)
            shapes[p + 'final.3.bias'] = (E,)
        shapes['emb_layer_norm_after.weight'] = (E,)
        shapes['emb_layer_norm_after.bias'] = (E,)
    elif kind == 'esmc':
        V, F = 64, swiglu_width(E)
        shapes['embed_tokens.weight'] = (V, E)
        for i in range(num_layers):
            p = f'layers.{i}.'
            shapes[p + 'self_attn.norm.weight'] = (E,)
            shapes[p + 'self_attn.norm.bias'] = (E,)
            for n in ('q', 'k', 'v', 'out'):
                shapes[p + f'self_attn.{n}.weight'] = (E, E)
            shapes[p + 'self_attn.layernorm_q.weight'] = (E,)
            shapes[p + 'self_attn.layernorm_k.weight'] = (E,)
            shapes[p + 'final.0.weight'] = (E,)
            shapes[p + 'final.0.bias'] = (E,)
            shapes[p + 'final.1.activation.weight'] = (F, E)
            shapes[p + 'final.1.fc.weight'] = (F, E)
            shapes[p + 'final.2.weight'] = (E, F)
        shapes['emb_layer_norm_after.weight'] = (E,)
    else:
        raise ValueError(f'unknown model kind {kind!r}')
    shapes['lm_head.dense.weight'] = (E, E)
    shapes['lm_head.dense.bias'] = (E,)
    shapes['lm_head.layer_norm.weight'] = (E,)
    shapes['lm_head.layer_norm.bias'] = (E,)
    shapes['lm_head.final.weight'] = (V, E)
    shapes['lm_head.final.bias'] = (V,)
    return shapes


def _is_norm_weight(name: str) -> bool:
    return name.endswith('.weight') and (
        'norm' in name or name.endswith('final.0.weight'))


def synthetic_tensor(name: str, shape: Tuple[int, ...], seed: int = 0,
                     sigma: float | None = None) -> torch.Tensor:
    """One bf16 tensor: LN weights 1+0.1n, biases 0.02n, matrices sigma*n.

    `sigma=None` picks 1/sqrt(fan_in) for matrices (keeps activations O(1)
    through depth so softmax/GELU see realistic ranges); embedding rows ~ n.
    """
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    a = rng.standard_normal(shape, dtype=np.float32)
    if len(shape) == 1:
        a = 1.0 + 0.1 * a if _is_norm_weight(name) else 0.02 * a
    elif name == 'embed_tokens.weight':
        a = a
    else:
        a *= (1.0 / math.sqrt(shape[1])) if sigma is None else sigma
    return torch.from_numpy(a).to(torch.bfloat16)


def synthetic_state_dict(kind: str, num_layers: int, embed_dim: int, seed: int = 0,
                         sigma: float | None = None) -> Dict[str, torch.Tensor]:
    return {n: synthetic_tensor(n, s, seed, sigma)
            for n, s in tensor_shapes(kind, num_layers, embed_dim).items()}


def massive_channel_state_dict(num_layers: int, embed_dim: int, scale: float, seed: int = 2, n_channels: int = 4, gain_layers=None, base=None):
    """The ill-conditioned ESM-2 probe model of tools/half_outlier_probe.py / bench.py (`precision_half.outlier_model`): the synthetic
    weights with `n_channels` residual-stream channels made MASSIVE -- their embedding columns and FFN-down biases multiplied by `scale`
    (so every layer feeds them again), the attention LayerNorm gains of the first two by min(scale, 10) (which pushes attention scores
    into the hundreds) -- the regime trained checkpoints are known for and N(0, 0.02) weights are not.  `gain_layers`: the layers whose gains are
    raised (default: all).  `base`: an already synthesised state dict of the same (num_layers, embed_dim, seed) -- the result then SHARES every
    untouched tensor with it and clones only the 1 + 2L patched ones (bench.py: no second 650 M-parameter synthesis).  Returns (state dict, channel ids)."""
    if base is None:
        w = synthetic_state_dict('esm2', num_layers, embed_dim, seed=seed)
    else:
        w = dict(base)
        for n in massive_channel_patched_names(num_layers):
            w[n] = base[n].clone()
    g = torch.Generator().manual_seed(0)
    cols = torch.randperm(embed_dim, generator=g)[:n_channels]
    w['embed_tokens.weight'][:, cols] *= scale
    for i in range(num_layers):
        w[f'layers.{i}.final.3.bias'][cols] *= scale
        if gain_layers is None or i in gain_layers:
            w[f'layers.{i}.self_attn.norm.weight'][cols[:2]] *= min(scale, 10.0)
    return w, cols


def massive_channel_patched_names(num_layers: int) -> List[str]:
    """The tensors massive_channel_state_dict changes (everything else equals the plain synthetic state dict of the same seed)."""
    names = ['embed_tokens.weight']
    for i in range(num_layers):
        names += [f'layers.{i}.final.3.bias', f'layers.{i}.self_attn.norm.weight']
    return names


def token_outlier_state_dict(kind: str, num_layers: int, embed_dim: int, scale: float, token_ids, seed: int = 2, n_channels: int = 4,
                             gain_scale: float = 10.0, base=None):
    """Counter-example model for the calibration of precision 'half' (VERDICT r5 item 1): the massive channels exist ONLY in the embedding
    rows of `token_ids` (e.g. `X`, `<unk>`, `<mask>`): those rows' entries in `n_channels` columns are set to +-`scale` (the typical entry is
    N(0, 1)), and the attention LayerNorm gains of the first two of these channels are multiplied by `gain_scale` in every layer.  A
    calibration batch that never contains these tokens sees a perfectly benign model.  Returns (state dict, channel ids)."""
    w = dict(base) if base is not None else synthetic_state_dict(kind, num_layers, embed_dim, seed=seed)
    g = torch.Generator().manual_seed(1)
    cols = torch.randperm(embed_dim, generator=g)[:n_channels]
    emb = w['embed_tokens.weight'].clone()
    for t in token_ids:
        sign = torch.where(torch.rand(n_channels, generator=g) < 0.5, -1.0, 1.0)
        emb[int(t), cols] = (sign * scale).to(emb.dtype)
    w['embed_tokens.weight'] = emb
    for i in range(num_layers):
        n = f'layers.{i}.self_attn.norm.weight'
        w[n] = w[n].clone()
        w[n][cols[:2]] *= gain_scale
    return w, cols


def checkpoint_metadata(name: str, num_layers: int, embed_dim: int, heads: int) -> Dict[str, str]:
    """The five metadata strings the loader reads (esm.py:329-339)."""
    return {'format': 'pt', 'name': name, 'num_layers': str(num_layers),
            'embed_dim': str(embed_dim), 'attention_heads': str(heads)}


def write_checkpoint(path: str, name: str, num_layers: int | None = None,
                     embed_dim: int | None = None, heads: int | None = None,
                     seed: int = 0, sigma: float | None = None) -> str:
    """Write a synthetic safetensors checkpoint in the reference layout."""
    from safetensors.torch import save_file
    kind = name.split('_')[0]
    if num_layers is None:
        kind, num_layers, embed_dim, heads = MODEL_ZOO[name]
    sd = synthetic_state_dict(kind, num_layers, embed_dim, seed, sigma)
    save_file(sd, path, metadata=checkpoint_metadata(name, num_layers, embed_dim, heads))
    return path


# ---------------------------------------------------------------- batches

def random_tokens(lengths: List[int], seed: int = 0) -> torch.Tensor:
    """Packed ids: `<cls>` + uniform residues 4..23 + `<eos>` per sequence."""
    rng = np.random.Generator(np.random.PCG64([seed, 0x70C]))
    parts = []
    for n in lengths:
        t = rng.integers(4, 24, size=n, dtype=np.int64)
        t[0], t[-1] = 0, 2
        parts.append(t)
    return torch.from_numpy(np.concatenate(parts))


def cu_lens_of(lengths: List[int]) -> torch.Tensor:
    cu = np.zeros(len(lengths) + 1, dtype=np.int32)
    np.cumsum(np.asarray(lengths, dtype=np.int64), out=cu[1:])
    return torch.from_numpy(cu)


def uniform_batch(total: int, seq_len: int, seed: int = 0):
    """`total` packed tokens as total//seq_len sequences of seq_len (+ a remainder)."""
    lengths = [seq_len] * (total // seq_len)
    if total % seq_len:
        lengths.append(total % seq_len)
    return random_tokens(lengths, seed), cu_lens_of(lengths), max(lengths), lengths


def proteome_lengths(total: int, seed: int = 0, median: float = 350.0, sigma: float = 0.75,
                     lo: int = 30, hi: int = 3500) -> List[int]:
    """Greedy token-budget packing of log-normal protein lengths (+2 for cls/eos).

    Sequences are appended until the next would exceed `total`
    (reference packing rule esme/data.py:42-51); the last is then shortened so
    the batch holds exactly `total` tokens.
    """
    rng = np.random.Generator(np.random.PCG64([seed, 0x9507]))
    lengths: List[int] = []
    used = 0
    while True:
        aa = int(np.clip(np.rint(rng.lognormal(math.log(median), sigma)), lo, hi)) + 2
        if used + aa > total:
            break
        lengths.append(aa)
        used += aa
    rest = total - used
    if rest >= 3:
        lengths.append(rest)
    elif rest > 0:
        lengths[-1] += rest
    return lengths


def proteome_batch(total: int, seed: int = 0):
    lengths = proteome_lengths(total, seed)
    return random_tokens(lengths, seed), cu_lens_of(lengths), max(lengths), lengths


def algorithmic_flops(kind: str, num_layers: int, embed_dim: int, lengths: List[int]) -> float:
    """T*[L(8E^2+cEF)+2E^2+2EV] + 4*E*L*sum(S_i^2)."""
    E, L = embed_dim, num_layers
    if kind == 'esm2':
        c, F, V = 4, 4 * E, 33
    else:
        c, F, V = 6, swiglu_width(E), 64
    T = sum(lengths)
    s2 = sum(s * s for s in lengths)
    return T * (L * (8 * E * E + c * E * F) + 2 * E * E + 2 * E * V) + 4.0 * E * L * s2
