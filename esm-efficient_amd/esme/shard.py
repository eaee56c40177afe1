"""Protein-sharded multi-GPU inference: the only place the path touches a collective.

The packed forward shards embarrassingly by protein (attention is block-diagonal over
sequences; the reference checks a sequence's logits do not depend on what it is packed
with, tests/test_esm.py:31-42).  The reference itself has no multi-GPU inference code
(SURVEY.md §2.2); this module is the MI355X-native plan of SURVEY.md §8(e):

  * one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in
    the CPU tests), weights replicated;
  * whole sequences are assigned to ranks, token-balanced (longest-first greedy into the
    least-loaded rank, ties broken on sum S^2 which is what attention costs);
  * each rank runs its own packed batch; ONE collective at the end all-gathers the
    (T_r, V) logits (padded to max T_r; 3.3 MB per rank at 50k residues x V=33) and rank
    order is undone with the inverse permutation so callers see input order.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np
import torch


def partition_sequences(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Indices of the sequences each rank owns.  Deterministic; every rank computes
    the same plan from the same lengths."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * world
    sq = [0] * world
    plan: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], sq[j], j))
        plan[r].append(i)
        load[r] += lengths[i]
        sq[r] += lengths[i] * lengths[i]
    for p in plan:
        p.sort()
    return plan


def local_batch(tokens: torch.Tensor, cu_lens: torch.Tensor, mine: Sequence[int]
                ) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """This rank's packed (tokens, cu_lens rebased to 0, max_len)."""
    cu = cu_lens.tolist()
    if not mine:
        return tokens[:0], torch.zeros(1, dtype=torch.int32), 0
    parts = [tokens[cu[i]:cu[i + 1]] for i in mine]
    lens = [cu[i + 1] - cu[i] for i in mine]
    local_cu = torch.zeros(len(mine) + 1, dtype=torch.int32)
    local_cu[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64), 0)
    return torch.cat(parts), local_cu, max(lens)


def gather_rows_all_ranks(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """All-gather row blocks of different heights: pad to max(counts), ONE
    all_gather_into_tensor, then strip the padding.  Returns the concatenation in rank
    order, (sum(counts), V)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    width = local.shape[1]
    tmax = max(counts)
    send = local
    if local.shape[0] != tmax:
        send = torch.zeros(tmax, width, dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    recv = torch.empty(world * tmax, width, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    return torch.cat([recv[r * tmax:r * tmax + counts[r]] for r in range(world)])


def _output_spec(forward, out_width, out_dtype):
    """(width, dtype) of `forward`'s rows for a rank that has no sequence to run.  Both are known locally -- from the arguments, or from
    the model when `forward` is a model or a bound method of one (vocab_size; fp32 logits in precision 'exact' / 'half', bf16
    otherwise) -- so no collective is spent on them."""
    model = forward if hasattr(forward, 'vocab_size') else getattr(forward, '__self__', None)
    if model is not None and hasattr(model, 'vocab_size'):
        if out_width is None:
            out_width = model.vocab_size
        if out_dtype is None:
            out_dtype = torch.float32 if getattr(model, 'precision', 'fast') in ('exact', 'half') else torch.bfloat16
    if out_width is None or out_dtype is None:
        raise ValueError('sharded_forward: a rank without sequences needs out_width / out_dtype (pass them, or pass the model / a bound model method)')
    return int(out_width), out_dtype


def sharded_forward(forward: Callable[[torch.Tensor, Tuple[torch.Tensor, int]], torch.Tensor],
                    tokens: torch.Tensor, cu_lens: torch.Tensor, device, group=None,
                    out_width: int = None, out_dtype=None) -> torch.Tensor:
    """Run `forward(tokens_r, (cu_lens_r, max_len_r))` on this rank's share of the packed
    batch and return the logits of the WHOLE batch, in input order, on every rank.  ONE collective: the logits all-gather.

    `tokens` / `cu_lens` are the full (host) batch, identical on all ranks.  `out_width` / `out_dtype`: shape of `forward`'s rows,
    needed only by a rank that receives no sequence (fewer sequences than ranks) when `forward` is not a model / bound model method.
    """
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    cu = cu_lens.tolist()
    lengths = [b - a for a, b in zip(cu[:-1], cu[1:])]
    plan = partition_sequences(lengths, world)
    counts = [sum(lengths[i] for i in p) for p in plan]
    tok_r, cu_r, max_r = local_batch(tokens, cu_lens, plan[rank])
    if tok_r.numel():
        out_r = forward(tok_r.to(device), (cu_r.to(device), max_r))
    else:
        out_r = None
    if out_r is None:                                     # width and dtype are known locally: no second collective
        width, dtype = _output_spec(forward, out_width, out_dtype)
        out_r = torch.zeros(0, width, dtype=dtype, device=device)
    gathered = gather_rows_all_ranks(out_r, counts, group)
    # gathered holds rank 0's sequences, then rank 1's, ...: build the inverse permutation
    src = np.concatenate([np.arange(cu[i], cu[i + 1]) for p in plan for i in p]) if len(lengths) else np.zeros(0, int)
    inv = torch.empty(len(src), dtype=torch.int64)
    inv[torch.from_numpy(src)] = torch.arange(len(src))
    return gathered[inv.to(gathered.device)]
