"""Varlen rotary position embedding on the HIP path.

Mirrors the reference's `RotaryEmbedding` (esme/rotary.py:81-165): fp32
inv_freq = base^(-2j/d), cos/sin tables of shape (max_len, d) (halves duplicated)
cached while max_len does not grow, cast to the activation dtype (bf16) -- but the
~10 ATen launches + host sync per call are one fused in-place kernel here, and
the row -> position map is computed once per forward instead of twice per layer.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from esme import _hip


def culen_indices(cu_lens: torch.Tensor) -> torch.Tensor:
    """Position of each packed row inside its sequence (reference rotary.py:5-14),
    int64 like the reference; computed by the HIP kernel (no host sync beyond the
    total length)."""
    total = int(cu_lens[-1])
    if cu_lens.is_cuda:
        return _hip.seq_positions(cu_lens, total)[0].to(torch.int64)
    cu = cu_lens.to(torch.int64)
    lengths = cu[1:] - cu[:-1]
    return torch.arange(total) - torch.repeat_interleave(cu[:-1], lengths)


class RotaryEmbedding(nn.Module):
    def __init__(self, dim: int, base: float = 10000.0, pos_idx_in_fp32: bool = True, device=None,
                 pad_to: Optional[int] = None):
        super().__init__()
        self.dim, self.base = dim, float(base)
        # tables `pad_to` wide for a head-padded layout: the real halves sit pad_to/2 apart, pad slots get
        # cos = 1, sin = 0 (esme.attention.head_slots)
        self.pad_to = pad_to if pad_to and pad_to != dim else None
        self._seq_len_cached = 0
        self._cos_cached: Optional[torch.Tensor] = None
        self._sin_cached: Optional[torch.Tensor] = None

    def _compute_inv_freq(self) -> torch.Tensor:
        return 1.0 / (self.base ** (torch.arange(0, self.dim, 2, dtype=torch.float32) / self.dim))

    def _update_cos_sin_cache(self, seqlen: int, device=None, dtype=torch.bfloat16) -> None:
        """Tables are built on the host in fp32 and rounded to `dtype` exactly as the
        reference does on its device (rotary.py:116-149), then uploaded once.  One cached pair per dtype (precision 'half' may need
        float16 tables for some layers and float32 tables for others in the same forward); `_cos_cached` / `_sin_cached` are the
        pair requested last."""
        cache = self.__dict__.setdefault('_table_cache', {})
        hit = cache.get(dtype)
        if hit is None or seqlen > hit[2] or hit[0].device != torch.device(device):
            seqlen = max(seqlen, hit[2] if hit is not None else 0, self._seq_len_cached)
            t = torch.arange(seqlen, dtype=torch.float32)
            ang = torch.outer(t, self._compute_inv_freq())
            ang = torch.cat((ang, ang), dim=-1)
            cos, sin = ang.cos().to(dtype), ang.sin().to(dtype)
            if self.pad_to is not None:
                h, hp = self.dim // 2, self.pad_to // 2
                pc, ps = torch.ones(seqlen, self.pad_to, dtype=dtype), torch.zeros(seqlen, self.pad_to, dtype=dtype)
                for src, dst in ((0, 0), (h, hp)):
                    pc[:, dst:dst + h] = cos[:, src:src + h]
                    ps[:, dst:dst + h] = sin[:, src:src + h]
                cos, sin = pc, ps
            hit = cache[dtype] = (cos.to(device), sin.to(device), seqlen)
            self._seq_len_cached = max(self._seq_len_cached, seqlen)
        self._cos_cached, self._sin_cached = hit[0], hit[1]

    def tables(self, max_len: int, device, dtype=torch.bfloat16) -> Tuple[torch.Tensor, torch.Tensor]:
        self._update_cos_sin_cache(max_len, device, dtype)
        return self._cos_cached, self._sin_cached

    def forward(self, q: torch.Tensor, k: torch.Tensor, cu_lens: torch.Tensor, max_len: int,
                pos: Optional[torch.Tensor] = None, inplace: bool = False):
        """q, k: (T, H, d) bf16.  Returns the rotated (q, k) as NEW tensors and leaves the inputs untouched, like the
        reference (esme/rotary.py:151-165).  `inplace=True` rotates the given tensors (views allowed if rows are
        contiguous) and returns them: what the layer stack does (esme/attention.py here), where nothing reuses the
        unrotated values and the copy would be a wasted HBM pass."""
        T, H, d = q.shape
        cos, sin = self.tables(max_len, q.device, q.dtype)
        if pos is None:
            pos, _ = _hip.seq_positions(cu_lens, T)
        if not inplace:
            q, k = q.clone(memory_format=torch.contiguous_format), k.clone(memory_format=torch.contiguous_format)
        _hip.rotary_(q.view(T, H * d), k.view(T, H * d), cos, sin, pos, H)
        return q, k
