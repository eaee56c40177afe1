"""Streaming inference over packed batches: host batching, H2D, forward and D2H overlapped.

The reference's inference driver is a plain loop (`workflow/inference/inference_on_human.py:56-58`:
`model(tokens.to(device), (cu_lens.to(device), max_len))` per `FastaTokenDataset` batch, results
pulled back synchronously).  Here the three stages run on three HIP streams chained by events:

    copy stream : pinned host tokens / cu_lens  --H2D-->  device
    compute     : forward (+ optional per-protein mean pooling) on the model's kernels
    output      : result --D2H--> pinned host buffer (a small ring), handed out when its event fired

so the next batch's upload and the previous batch's download ride under the current forward, and the
Python loop never blocks on the GPU except when the ring is full.  Results are yielded in input order.
"""
from __future__ import annotations

from collections import deque
from typing import Iterable, Iterator, Optional, Tuple

import torch

from esme.pooling import partition_mean_pool


class _PinnedRing:
    """`slots` reusable pinned host buffers (allocating pinned memory per batch would cost more than the copy)."""

    def __init__(self, slots: int):
        self.bufs = [None] * slots
        self.i = 0

    def take(self, shape, dtype) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        n = max(n, 1)
        buf = self.bufs[self.i]
        if buf is None or buf.dtype != dtype or buf.numel() < n:
            buf = torch.empty(n, dtype=dtype, pin_memory=True)
            self.bufs[self.i] = buf
        self.i = (self.i + 1) % len(self.bufs)
        return buf[:n].view(*shape)


class StreamedInference:
    """`for out in StreamedInference(model, ...).run(batches)` with `batches` yielding
    `(tokens int64 (T,), (cu_lens int32 (B+1,), max_len))` on the host, e.g. an `esme.data.FastaTokenDataset`
    or its DataLoader.  `what` is 'forward' (logits), 'predict_log_prob' or 'forward_representation';
    `pool='mean'` reduces a representation to one row per protein on the GPU before the download.
    Each result is a pinned host tensor that stays valid until `depth + 1` further results were produced
    (clone it to keep it longer)."""

    def __init__(self, model, what: str = 'forward_representation', pool: Optional[str] = None, depth: int = 2):
        assert what in ('forward', 'predict_log_prob', 'forward_representation')
        assert pool in (None, 'mean')
        assert pool is None or what == 'forward_representation', 'pooling applies to representations'
        self.model, self.fn, self.pool, self.depth = model, getattr(model, what), pool, max(1, depth)
        self.device = model.embed_tokens.weight.device

    def run(self, batches: Iterable[Tuple[torch.Tensor, Tuple[torch.Tensor, int]]]) -> Iterator[torch.Tensor]:
        dev = self.device
        compute = torch.cuda.current_stream(dev)
        copy_s, out_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ring_in, ring_cu = _PinnedRing(self.depth + 2), _PinnedRing(self.depth + 2)
        # result i is handed out in iteration i + depth and must stay valid until depth + 1 further results were
        # produced, i.e. through iteration i + 2*depth + 1: its slot may be re-targeted in iteration i + 2*depth + 2
        ring_out = _PinnedRing(2 * self.depth + 2)
        pending = deque()
        with torch.no_grad():
            for tokens, (cu_lens, max_len) in batches:
                tok_h = ring_in.take(tokens.shape, torch.int64)
                tok_h.copy_(tokens)
                cu_h = ring_cu.take(cu_lens.shape, torch.int32)
                cu_h.copy_(cu_lens)
                with torch.cuda.stream(copy_s):
                    tok_d = tok_h.to(dev, non_blocking=True)
                    cu_d = cu_h.to(dev, non_blocking=True)
                    uploaded = torch.cuda.Event()
                    uploaded.record(copy_s)
                compute.wait_event(uploaded)
                tok_d.record_stream(compute)                  # allocated on the copy stream, consumed on compute
                cu_d.record_stream(compute)
                out = self.fn(tok_d, (cu_d, int(max_len)))
                if self.pool == 'mean':
                    out = partition_mean_pool(out, cu_d)
                computed = torch.cuda.Event()
                computed.record(compute)
                host = ring_out.take(out.shape, out.dtype)
                with torch.cuda.stream(out_s):
                    out_s.wait_event(computed)
                    host.copy_(out, non_blocking=True)
                    out.record_stream(out_s)
                    downloaded = torch.cuda.Event()
                    downloaded.record(out_s)
                pending.append((host, downloaded))
                while len(pending) > self.depth:
                    h, ev = pending.popleft()
                    ev.synchronize()
                    yield h
            while pending:
                h, ev = pending.popleft()
                ev.synchronize()
                yield h
