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
    (clone it to keep it longer).

    Feeding it: a DataLoader with `num_workers=0` is the fast choice on one GPU -- reading, tokenising and packing a 50 000-token batch takes ~20 ms
    of one core against 60 - 70 ms of GPU time, and three batches are in flight.  Worker PROCESSES forked from a process that already holds a GPU
    context stall its queues for 1 - 3 s when they start (copy-on-write protection of the parent's pages makes the driver evict and restore the
    queues: profiles/r05_e2e_fork_stall.txt); if workers are needed, create the DataLoader's iterator before the model is loaded, or use the
    'forkserver' start method."""

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
        # precision 'half': the run-time range guard (ESM2.check_overflow) would cost a device synchronisation per batch if predict_* asked for
        # it inline.  Here the sticky device flag travels to the host WITH each result (4 bytes on the output stream, pinned) and is looked at
        # when the result is handed out: no extra synchronisation, and an activation that left fp16's range surfaces as OverflowError at the
        # first result that may carry it.
        # The plan guard travels the same way (round 6): ESM2._guard_snapshot() judges the batch's device maxima against the plan ON the device, the
        # few KB of the verdict vector ride to the host with the result, and a stale plan is widened when the first result computed under it is
        # handed out (RuntimeWarning naming the batch; later batches run covered; results already in flight are named too).
        guard = getattr(self.model, 'precision', None) == 'half'
        ring_flag = _PinnedRing(2 * self.depth + 2) if guard else None
        self._ring_plan = _PinnedRing(2 * self.depth + 2) if guard else None
        deferred_before = getattr(self.model, '_defer_overflow', False)
        if guard:
            self.model._defer_overflow = True
        try:
            yield from self._run(batches, dev, compute, copy_s, out_s, ring_in, ring_cu, ring_out, ring_flag, pending)
        finally:
            if guard:
                self.model._defer_overflow = deferred_before

    def _hand_out(self, item, index):
        h, ev, flag_h, plan_h = item
        ev.synchronize()
        if plan_h is not None and float(plan_h[0]) != 0.0:
            if self.model._plan_verdict(plan_h.clone(), update=True, where=f' (batch {index} of this stream)') is None:
                import warnings                  # (an earlier batch's verdict has widened the plan meanwhile; THIS result still predates that)
                warnings.warn(f"precision='half': batch {index} of this stream was computed under the plan an earlier batch showed to be stale (it has been "
                              "widened since); re-run it if the mode's 1e-3 must hold for it", RuntimeWarning, stacklevel=2)
        if flag_h is not None and int(flag_h[0]) != 0:
            self.model._overflow_flag(self.device).zero_()
            raise OverflowError(f"precision='half': an activation left IEEE fp16's range (|x| >= 65 504) by batch {index} of this stream; its result "
                                "(and possibly the next batch's) holds inf / NaN.  Use precision 'exact' for this checkpoint / input.")
        return h

    def _run(self, batches, dev, compute, copy_s, out_s, ring_in, ring_cu, ring_out, ring_flag, pending):
        handed = 0
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
                host = ring_out.take(out.shape, out.dtype)
                snap = self.model._guard_snapshot() if ring_flag is not None else None      # (compute stream, before `computed`: judged against the plan this batch ran with)
                computed = torch.cuda.Event()
                computed.record(compute)
                flag_h = ring_flag.take((1,), torch.int32) if ring_flag is not None else None
                plan_h = self._ring_plan.take(snap.shape, torch.float32) if snap is not None else None
                with torch.cuda.stream(out_s):
                    out_s.wait_event(computed)
                    host.copy_(out, non_blocking=True)
                    out.record_stream(out_s)
                    if flag_h is not None:
                        flag_h.copy_(self.model._overflow_flag(dev), non_blocking=True)
                    if plan_h is not None:
                        plan_h.copy_(snap, non_blocking=True)
                        snap.record_stream(out_s)
                    downloaded = torch.cuda.Event()
                    downloaded.record(out_s)
                pending.append((host, downloaded, flag_h, plan_h))
                while len(pending) > self.depth:
                    yield self._hand_out(pending.popleft(), handed)
                    handed += 1
            while pending:
                yield self._hand_out(pending.popleft(), handed)
                handed += 1
