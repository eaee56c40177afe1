#!/usr/bin/env python
"""Tile-walk (raster) sweep for the four production GEMMs of an ESM2-650M layer, WITH their fused epilogues: band height gm x
group width gn through the per-call options (esme_gemm_opts_t.raster_*), interleaved rounds, median per setting."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import _hip

T, E, H = int(os.environ.get('T', 50000)), 1280, 20
d = E // H
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E); h4 = bf(T, 4 * E)
wqkv = bf(3 * E, E, scale=E ** -0.5)
wo, bo = bf(E, E, scale=E ** -0.5), bf(E, scale=0.1)
w1 = bf(4 * E, E, scale=E ** -0.5)
w2, b2 = bf(E, 4 * E, scale=(4 * E) ** -0.5), bf(E, scale=0.1)
NB = _hip.stats_blocks(T, E)
stats = (_hip.row_sums(x) / NB).expand(NB, T, 2).contiguous()
c1q, c2q = torch.randn(3 * E, device=dev), torch.randn(3 * E, device=dev)
c11, c21 = torch.randn(4 * E, device=dev), torch.randn(4 * E, device=dev)
partial = torch.empty(NB, T, 2, device=dev)
pos = (torch.arange(T, device=dev, dtype=torch.int32) % 500).contiguous()
ang = torch.outer(torch.arange(500.), 1.0 / (10000 ** (torch.arange(0, d, 2) / d)))
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos().to(torch.bfloat16).to(dev), ang.sin().to(torch.bfloat16).to(dev)
qkv = torch.empty(T, 3 * E, device=dev, dtype=torch.bfloat16)
u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16)
y = x.clone()
rot = (cos, sin, pos, d, 2 * E)
gemms = {
    'qkv +rot+lnf (15 cols)':    lambda: _hip.gemm_fused(x, wqkv, None, out=qkv, rot=rot, ln=(stats, E, 1e-5, c1q, c2q)),
    'out resid+stats (5 cols)':  lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=partial),
    'ffn1 gelu+lnf (20 cols)':   lambda: _hip.gemm_fused(x, w1, None, _hip.EPI_GELU, out=u, ln=(stats, E, 1e-5, c11, c21)),
    'ffn2 resid+stats (5 cols)': lambda: _hip.gemm_fused(h4, w2, b2, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=partial),
}
rasters = [None] + [(gm, gn) for gm in (1, 2, 4, 6, 8, 12, 16, 32) for gn in (1, 2, 3, 4, 5, 8, 10, 20)]
if os.environ.get('RASTERS'):      # e.g. RASTERS='4,3;6,2;2,5'
    rasters = [None] + [tuple(int(v) for v in r.split(',')) for r in os.environ['RASTERS'].split(';')]
ROUNDS, ITERS = int(os.environ.get('ROUNDS', 3)), int(os.environ.get('ITERS', 10))
for name, fn in gemms.items():
    ncol = int(name.split('(')[1].split()[0])
    rs = [r for r in rasters if r is None or r[1] <= ncol]
    times = {r: [] for r in rs}
    for _ in range(ROUNDS):
        for r in rs:
            ctx = _hip.gemm_options(raster=r) if r else _hip.gemm_options()
            with ctx:
                fn(); fn()
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(ITERS): fn()
                en.record(); torch.cuda.synchronize()
            times[r].append(st.elapsed_time(en) / ITERS * 1e3)
    base = statistics.median(times[None])
    ranked = sorted(((statistics.median(t), r) for r, t in times.items()), key=lambda z: z[0])
    print(f'{name}: default {base:.1f} us; best five: ' + ', '.join(f'{r}: {m:.1f} ({(m / base - 1) * 100:+.1f} %)' for m, r in ranked[:5])
          + '; worst: ' + ', '.join(f'{r}: {m:.1f}' for m, r in ranked[-2:]), flush=True)
