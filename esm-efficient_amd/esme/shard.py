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


_LOGIT_METHODS = ('forward', '__call__', 'predict_log_prob', 'predict_prob')


def _output_spec(forward, out_width, out_dtype):
    """(width, dtype) of `forward`'s rows for a rank that has no sequence to run, or None when they cannot be PROVEN locally.  They are
    known locally from the arguments, or from the model when `forward` is a model or one of its LOGIT-returning bound methods
    (`forward` / `predict_log_prob` / `predict_prob`: vocab_size wide; fp32 in precision 'exact' / 'half', bf16 otherwise).  Any other
    callable (`model.forward_representation`, a lambda around it, ...) has rows this function knows nothing about: the caller then
    agrees on them with one small collective (a (0, vocab) block against (tmax, embed_dim) blocks would hang the gather)."""
    if out_width is not None and out_dtype is not None:
        return int(out_width), out_dtype
    model = None
    if hasattr(forward, 'vocab_size') and hasattr(forward, 'forward_representation'):
        model = forward                                       # the model itself: model(...) returns logits
    elif getattr(forward, '__name__', None) in _LOGIT_METHODS and hasattr(getattr(forward, '__self__', None), 'vocab_size'):
        model = forward.__self__
    if model is None:
        return None
    width = model.vocab_size if out_width is None else out_width
    dtype = out_dtype if out_dtype is not None else (
        torch.float32 if getattr(model, 'precision', 'fast') in ('exact', 'half') else torch.bfloat16)
    return int(width), dtype


_DTYPE_CODES = (torch.bfloat16, torch.float32, torch.float16, torch.float64)


def _agree_output_spec(out_r, device, group):
    """One MAX all-reduce of (width, dtype code) for callables whose row shape is not known locally (see _output_spec): ranks that ran
    contribute their output's, idle ranks zeros.  Only taken when some rank is idle AND the spec is unknown -- every rank takes the same
    branch because both conditions are functions of the (identical) plan and arguments."""
    import torch.distributed as dist
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if out_r is not None:
        meta[0], meta[1] = out_r.shape[1], _DTYPE_CODES.index(out_r.dtype) + 1
    dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
    width, code = int(meta[0]), int(meta[1])
    if code == 0:
        raise ValueError('sharded_forward: no rank produced an output (empty batch) and out_width / out_dtype were not given')
    return width, _DTYPE_CODES[code - 1]


def sharded_forward(forward: Callable[[torch.Tensor, Tuple[torch.Tensor, int]], torch.Tensor],
                    tokens: torch.Tensor, cu_lens: torch.Tensor, device, group=None,
                    out_width: int = None, out_dtype=None) -> torch.Tensor:
    """Run `forward(tokens_r, (cu_lens_r, max_len_r))` on this rank's share of the packed
    batch and return the logits of the WHOLE batch, in input order, on every rank.  ONE collective: the logits all-gather.

    `tokens` / `cu_lens` are the full (host) batch, identical on all ranks.  `out_width` / `out_dtype`: shape of `forward`'s rows,
    used only when some rank receives no sequence (fewer sequences than ranks); unnecessary for a model or its logit methods
    (`model`, `model.forward`, `model.predict_log_prob`, `model.predict_prob`), and for any other callable their absence costs one extra
    16-byte all-reduce in that case instead of a wrong guess.
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
    if any(c == 0 for c in counts):                       # some rank is idle (every rank sees the same plan)
        spec = _output_spec(forward, out_width, out_dtype)        # a function of the arguments only: the same answer on every rank
        if spec is None:                                  # rows of an arbitrary callable: agree with one tiny collective (never for logits)
            spec = _agree_output_spec(out_r, device, group)
        if out_r is None:
            out_r = torch.zeros(0, spec[0], dtype=spec[1], device=device)
    gathered = gather_rows_all_ranks(out_r, counts, group)
    # gathered holds rank 0's sequences, then rank 1's, ...: build the inverse permutation
    src = np.concatenate([np.arange(cu[i], cu[i + 1]) for p in plan for i in p]) if len(lengths) else np.zeros(0, int)
    inv = torch.empty(len(src), dtype=torch.int64)
    inv[torch.from_numpy(src)] = torch.arange(len(src))
    return gathered[inv.to(gathered.device)]
