#!/usr/bin/env python
"""Time per round of the production GEMMs as a function of how many tiles a persistent workgroup walks (1, 2, 3, ... rounds of the 256 CUs), persistent
against one tile per workgroup, interleaved on one box.  Round 6: the pair-stream residual kernel showed a loss that appears only from the THIRD tile on
(profiles/r06_half_guard_regression.txt) -- do the bf16 fast-path kernels hide one too?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import _hip
torch.manual_seed(0)
dev = 'cuda'
E = 1280
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
def timed(fn, iters=20):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
shapes = {'out-proj resid+stats (N 1280, K 1280)': (E, E, 'resid'), 'FFN-down resid+stats (N 1280, K 5120)': (E, 4 * E, 'resid'),
          'FFN-up gelu+lnf (N 5120, K 1280)': (4 * E, E, 'ffn1')}
for name, (N, K, kind) in shapes.items():
    tn = N // 256
    print(f'== {name}')
    for rounds in [float(r) for r in os.environ.get('ROUNDS', '1,2,3,4,5,6,8,12').split(',')]:
        tm = int(rounds * 256) // tn                  # row tiles so that tiles <= rounds * 256
        M = tm * 256
        x = bf(M, K); w = bf(N, K, scale=K ** -0.5); b = bf(N, scale=0.1)
        if kind == 'resid':
            y = bf(M, N); part = torch.empty(_hip.stats_blocks(M, N), M, 2, device=dev)
            fn = lambda: _hip.gemm_fused(x, w, b, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=part)
        else:
            xs = bf(M, E); NB = _hip.stats_blocks(M, E)
            stats = (_hip.row_sums(xs) / NB).expand(NB, M, 2).contiguous()
            c1, c2 = torch.randn(N, device=dev), torch.randn(N, device=dev)
            u = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            fn = lambda: _hip.gemm_fused(xs, w, None, _hip.EPI_GELU, out=u, ln=(stats, E, 1e-5, c1, c2))
        t = {0: [], 1: []}
        for _ in range(3):
            for p in (1, 0):
                with _hip.gemm_options(persist=p):
                    t[p].append(timed(fn))
        a, c = statistics.median(t[1]), statistics.median(t[0])
        tiles = tm * tn
        print(f'  {tiles:5d} tiles = {tiles / 256:5.2f} rounds  M = {M:6d}: persistent {a:7.1f} us ({a / (tiles / 256):6.1f} per round)   one tile per workgroup {c:7.1f} us ({c / (tiles / 256):6.1f} per round)   {100 * (a / c - 1):+.1f} %')
