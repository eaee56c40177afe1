"""Learned positional embedding of ESM-1b / ESM-1v.

Same name, constructor and index helpers as the reference (`esme/embedding.py:7-107`):
a (max_positions + 2, E) table whose row `padding_idx` (= 1) belongs to `<pad>`; position
p (1-based inside a sequence) reads row p + padding_idx.  The index helpers are host-side
integer code; the table lookup itself is fused with the token lookup in one HIP kernel
(`esme_hip_embed_positions`), see `ESM1b.embedding`.
"""
from __future__ import annotations

import torch
from torch import nn

from esme.alphabet import Alphabet


class LearnedPositionalEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, dtype=torch.bfloat16):
        super().__init__()
        self.padding_idx = Alphabet.padding_idx
        self.max_positions = num_embeddings
        self.num_embeddings = num_embeddings + 2
        self.embedding_dim = embedding_dim
        self.weight = nn.Parameter(torch.zeros(self.num_embeddings, embedding_dim, dtype=dtype), requires_grad=False)

    def positions(self, input: torch.Tensor) -> torch.Tensor:
        """(B, S) ids -> row indices: running count of non-pad tokens + padding_idx, pads -> padding_idx."""
        if input.size(1) > self.max_positions:
            raise ValueError(f'Sequence length {input.size(1)} above maximum  sequence length of {self.max_positions}')
        keep = input.ne(self.padding_idx).int()
        return (torch.cumsum(keep, dim=1).type_as(keep) * keep).long() + self.padding_idx

    def position_unpad(self, input: torch.Tensor, pad_args) -> torch.Tensor:
        """Packed (T,) ids + (cu_lens, max_len) -> row indices 2, 3, ... restarting at every sequence."""
        assert input.ndim == 1
        cu_lens, max_len = pad_args
        if max_len > self.max_positions:
            raise ValueError(f'Sequence length {max_len} above maximum  sequence length of {self.max_positions}')
        cu = cu_lens.to(torch.int64)
        lens = cu[1:] - cu[:-1]
        start = torch.repeat_interleave(cu[:-1], lens)
        return torch.arange(int(cu[-1]), device=input.device) - start + 1 + self.padding_idx

    def forward(self, input: torch.Tensor, pad_args=None) -> torch.Tensor:
        """Position rows only (stand-alone use; the model path fuses this with the token lookup)."""
        from esme import _hip
        idx = self.positions(input) if pad_args is None else self.position_unpad(input, pad_args)
        return _hip.gather_rows(self.weight, idx.reshape(-1).to(self.weight.device)).view(*input.shape, -1)
