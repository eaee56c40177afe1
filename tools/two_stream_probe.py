#!/usr/bin/env python
"""Does filling one kernel's tail / prologue with another stream's work help at the power cap?  The four
GEMMs of an ESM2-650M layer on 50 000 rows in one stream vs two half-batches on two streams."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import _hip
T, E = 50000, 1280
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E)
wqkv, wo, w1, w2 = bf(3 * E, E, scale=E ** -0.5), bf(E, E, scale=E ** -0.5), bf(4 * E, E, scale=E ** -0.5), bf(E, 4 * E, scale=(4 * E) ** -0.5)
b1, bo = bf(4 * E, scale=0.1), bf(E, scale=0.1)
def layer(xs, bufs):
    qkv, u, y = bufs
    _hip.gemm_fused(xs, wqkv, None, out=qkv)
    _hip.gemm_fused(qkv[:, :E], wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y)
    _hip.gemm_fused(y, w1, b1, _hip.EPI_GELU, out=u)
    _hip.gemm_fused(u, w2, bo, _hip.EPI_RESIDUAL, y, 1.0, y)
def bufs(n): return (torch.empty(n, 3 * E, device=dev, dtype=torch.bfloat16), torch.empty(n, 4 * E, device=dev, dtype=torch.bfloat16), bf(n, E))
full = bufs(T)
h = T // 2
halves = [(x[:h], bufs(h)), (x[h:], bufs(T - h))]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one_stream():
    for _ in range(4): layer(x, full)
def two_streams():
    cur = torch.cuda.current_stream()
    for s, (xs, b) in zip((s1, s2), halves):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(4): layer(xs, b)
    cur.wait_stream(s1); cur.wait_stream(s2)
def seq_halves():
    for xs, b in halves:
        for _ in range(4): layer(xs, b)
res = {k: [] for k in ('one stream, M=50000', 'two streams, 2 x M=25000', 'one stream, 2 x M=25000')}
for r in range(5):
    for name, fn in zip(res, (one_stream, two_streams, seq_halves)):
        fn(); torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); fn(); fn(); en.record(); torch.cuda.synchronize()
        res[name].append(st.elapsed_time(en) / 8)
for k, v in res.items():
    print(f'{k:28s} {statistics.median(v) * 1e3:8.1f} us per layer-equivalent', flush=True)
