"""hipGraph replay of the packed forward for a fixed input shape.

A forward of a 30-layer model is ~160 kernel launches through ctypes; below ~10 ms of GPU work the
host cannot enqueue them fast enough (measured: ESM2-150M on 8 192 residues needs 6 ms of GPU time
and 9 ms of Python).  All launches go to the current HIP stream with caller-owned buffers, so a
forward with a fixed `(T, n_sequences, max_len)` can be stream-captured once into a hipGraph
(`torch.cuda.CUDAGraph`, which on ROCm records hipGraph nodes) and replayed with one launch:
`tokens` and `cu_lens` live in static device buffers that are overwritten before each replay.

The shape key includes `max_len` (it sets the attention grid) and the number of sequences (it
sizes `cu_lens`); batches of masked copies of one protein (`esme.variant`) and fixed-size
benchmark batches repeat their shape, token-budget FASTA batches generally do not -- those stay on
the eager path.
"""
from __future__ import annotations

from collections import OrderedDict

import torch


class GraphedForward:
    def __init__(self, model, what: str, n_tokens: int, n_seqs: int, max_len: int, device):
        self.fn = getattr(model, what)
        self.max_len = int(max_len)
        self.tokens = torch.zeros(n_tokens, dtype=torch.int64, device=device)
        self.cu_lens = torch.zeros(n_seqs + 1, dtype=torch.int32, device=device)
        self.graph = None
        self.out = None

    def _capture(self):
        side = torch.cuda.Stream(device=self.tokens.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):          # builds packed / folded weights, rotary tables, kernel attributes
                self.fn(self.tokens, (self.cu_lens, self.max_len))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            self.out = self.fn(self.tokens, (self.cu_lens, self.max_len))
        self.graph = graph

    def run(self, tokens: torch.Tensor, cu_lens: torch.Tensor, clone: bool = True) -> torch.Tensor:
        self.tokens.copy_(tokens, non_blocking=True)
        self.cu_lens.copy_(cu_lens, non_blocking=True)
        if self.graph is None:
            self._capture()
        self.graph.replay()
        return self.out.clone() if clone else self.out


class GraphCache:
    """Per-model LRU of captured forwards, keyed by (method, T, n_sequences, max_len)."""

    def __init__(self, model, capacity: int = 8):
        self.model, self.capacity = model, capacity
        self.entries: 'OrderedDict[tuple, GraphedForward]' = OrderedDict()

    def run(self, what: str, tokens: torch.Tensor, pad_args, clone: bool = True) -> torch.Tensor:
        cu_lens, max_len = pad_args
        assert tokens.ndim == 1, 'graph replay serves the packed (1-D tokens) path'
        key = (what, tokens.numel(), cu_lens.numel() - 1, int(max_len), str(tokens.device))
        g = self.entries.get(key)
        if g is None:
            g = GraphedForward(self.model, what, tokens.numel(), cu_lens.numel() - 1, int(max_len), tokens.device)
            self.entries[key] = g
            while len(self.entries) > self.capacity:
                self.entries.popitem(last=False)
        else:
            self.entries.move_to_end(key)
        return g.run(tokens, cu_lens, clone)
