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

# attributes under which modules cache DERIVED device tensors (packed / LayerNorm-folded / padded weight copies,
# rotary tables): allocated outside a graph's private pool, read by the captured kernels through raw pointers
_DERIVED_ATTRS = ('_qkv_w', '_qkv_b', '_fold', '_out_w', '_out_b', '_down_pad', '_packed', '_pad', '_embed_pad',
                  '_cos_cached', '_sin_cached', '_fold16', '_out16', '_down16', '_rho16', '_fold16x', '_half_ovf', '_up_pad', '_table_cache')


def _flatten(x):
    if isinstance(x, torch.Tensor):
        yield x
    elif isinstance(x, (tuple, list)):
        for y in x:
            yield from _flatten(y)
    elif isinstance(x, dict):
        for y in x.values():
            yield from _flatten(y)


def external_tensors(model):
    """Every derived tensor a forward of `model` reads that lives outside the capture pool.  A captured graph
    keeps strong references to them: modules REPLACE such tensors (the rotary cache regrows when a longer
    batch arrives; weight copies are rebuilt after a parameter update) and the replaced ones would otherwise
    be freed while an older graph still holds their addresses."""
    keep = []
    for m in model.modules():
        for a in _DERIVED_ATTRS:
            keep.extend(_flatten(getattr(m, a, None)))
    keep.extend(p.data for p in model.parameters())
    plan = getattr(model, '_half_plan', None)             # precision 'half': the massive-channel list the extension K-tile kernels read
    if plan is not None and getattr(plan, 'ext_sel', None) is not None:
        keep.append(plan.ext_sel)
    guard = getattr(model, '_half_guard', None)           # ... and the plan guard's device maxima / the range flag the captured kernels write
    if guard is not None:
        keep += [guard.col, guard.qk]
    if getattr(model, '_half_ovf', None) is not None:
        keep.append(model._half_ovf)
    return keep


class GraphedForward:
    def __init__(self, model, what: str, n_tokens: int, n_seqs: int, max_len: int, device):
        self.fn = getattr(model, what)
        self.max_len = int(max_len)
        self.tokens = torch.zeros(n_tokens, dtype=torch.int64, device=device)
        self.cu_lens = torch.zeros(n_seqs + 1, dtype=torch.int32, device=device)
        self.model = model
        self.graph = None
        self.out = None
        self._keep = None           # strong references to everything captured from outside the graph's pool

    def _capture(self):
        side = torch.cuda.Stream(device=self.tokens.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):          # builds packed / folded weights, rotary tables, kernel attributes
                self.fn(self.tokens, (self.cu_lens, self.max_len))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        guard = getattr(self.model, '_half_guard', None)
        if guard is not None:
            guard.clear()               # the warm-up forwards ran on placeholder tokens (all id 0): not data the plan should be held to
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            self.out = self.fn(self.tokens, (self.cu_lens, self.max_len))
        self._keep = external_tensors(self.model)      # e.g. the rotary tables of THIS max_len survive a later regrow
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

    def clear(self):
        """Drop every captured graph (call after changing weights: a graph replays the weight copies it was
        captured with -- memory-safe because it keeps them alive, but stale)."""
        self.entries.clear()
