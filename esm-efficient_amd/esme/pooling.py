"""Per-protein pooling of packed residue embeddings.

`partition_mean_pool` / `PartitionMeanPool` keep the reference's names and argument
order (`esme/pooling.py:8-69`): the mean of the rows `cu_lens[i] : cu_lens[i+1]` of a
packed (T, E) embedding for every protein i.  On MI355X it is one segmented-reduction
kernel (`esme_hip_segment_mean`): each protein's rows are read once with 16-byte loads
and accumulated in fp32 -- the reference accumulates in the embedding dtype with
`index_add_`, so for bf16 inputs this path is the more accurate of the two.

The attention-pooling heads of the reference (`AttentionPool`, `LearnedAggregation`, ...,
esme/pooling.py:72-238) are trainable task heads outside the inference hot path
(SURVEY.md §8 out-of-scope).
"""
from __future__ import annotations

import torch
from torch import nn

from esme import _hip


def partition_mean_pool(embed: torch.Tensor, cu_lens: torch.Tensor) -> torch.Tensor:
    """(B, E) means of the packed rows of `embed` (T, E) bf16 / fp32 on a HIP device."""
    return _hip.segment_mean(embed, cu_lens)


class PartitionMeanPool(nn.Module):
    def forward(self, embed, cu_lens):
        return partition_mean_pool(embed, cu_lens)

    @staticmethod
    def _indices(cu_lens):
        """Protein index of every packed row (reference esme/pooling.py:30-36)."""
        lens = (cu_lens[1:] - cu_lens[:-1]).to(torch.long)
        return torch.repeat_interleave(torch.arange(lens.numel(), device=cu_lens.device), lens)
