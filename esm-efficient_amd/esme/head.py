"""LM head on the HIP path (reference esme/head.py:8-27: dense -> GELU -> LN -> vocab).

The exact-erf GELU is the dense GEMM's epilogue; the vocab projection (N = 33 / 64,
HBM-bound) runs on the same GEMM kernel with row-clamped weight loads."""
from __future__ import annotations

import torch
from torch import nn

from esme import _hip
from esme.nn import LayerNorm, Linear


class RobertaLMHead(nn.Module):
    def __init__(self, embed_dim, vocab_size, dtype=torch.bfloat16):
        super().__init__()
        self.dense = Linear(embed_dim, embed_dim, dtype=dtype)
        self.layer_norm = LayerNorm(embed_dim, dtype=dtype)
        self.final = Linear(embed_dim, vocab_size, dtype=dtype)

    def forward(self, features):
        shape = features.shape
        x = features.reshape(-1, shape[-1])
        h = self.dense(x, _hip.EPI_GELU)
        self.layer_norm(h, out=h)
        y = self.final(h)
        return y.view(*shape[:-1], y.shape[-1])
