#!/usr/bin/env python
"""Column split of residual GEMMs whose width ends in a half-empty 256-column tile (ESMC-600M: N = 1 152): one 256 x 256 launch (esme_gemm_opts_t.tile = 2)
against full tiles + a 128 x 128 launch on the last 128 columns (the heuristic since round 6), interleaved on one box; bf16 and the fp16 pair stream."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'esm-efficient_amd')):
    sys.path.insert(0, p)
import torch
from esme import _hip
torch.manual_seed(0)
dev = 'cuda'
M = int(os.environ.get('M', 32064))


def timed(fn, iters=20):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for N, K, name in ((1152, 1152, 'out-projection'), (1152, 3072, 'FFN down')):
    x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    st = torch.empty(_hip.stats_blocks(M, N), M, 2, dtype=torch.float32, device=dev)
    x16, w16 = x.to(torch.float16), w.to(torch.float16)
    xs = torch.randn(M, 2 * N, device=dev).to(torch.float16)
    rho = (0.71 + 0.7 * torch.rand(N, device=dev))
    cases = {}
    for tile, label in ((2, 'one 256 x 256 launch'), (0, 'full tiles + 128-column launch')):
        def bf(tile=tile):
            with _hip.gemm_options(tile=tile):
                _hip.gemm_fused(x, w, None, _hip.EPI_RESIDUAL, res, 0.5, out=res, stats_out=st)
        def pair(tile=tile):
            with _hip.gemm_options(tile=tile):
                _hip.gemm_fused(x16, w16, None, _hip.EPI_RESIDUAL, None, 0.5, stats_out=st, resid_pair=xs, pair_scale=(rho, rho))
        cases[f'bf16, {label}'] = bf
        cases[f'fp16 pair stream, {label}'] = pair
    t = {k: [] for k in cases}
    for _ in range(5):
        for k, f in cases.items():
            t[k].append(timed(f))
    print(f'{name}: M = {M}, N = {N}, K = {K}')
    for k in cases:
        print(f'  {k:<52s} {statistics.median(t[k]):7.1f} us')
