#!/usr/bin/env python
"""Persistent 256 x 256 workgroups (esme_gemm_opts_t.persist) vs one workgroup per tile: bit equality of the four
production GEMMs of an ESM2-650M layer with their fused epilogues (ragged M), and interleaved timing."""
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
lib = _hip.load()
x = bf(T, E); h4 = bf(T, 4 * E)
wqkv = bf(3 * E, E, scale=E ** -0.5); wo, bo = bf(E, E, scale=E ** -0.5), bf(E, scale=0.1)
w1 = bf(4 * E, E, scale=E ** -0.5); w2, b2 = bf(E, 4 * E, scale=(4 * E) ** -0.5), bf(E, scale=0.1)
stats1 = _hip.row_sums(x)
NB = _hip.stats_blocks(T, E)
stats = (stats1 / NB).expand(NB, T, 2).contiguous()
c1q, c2q = torch.randn(3 * E, device=dev), torch.randn(3 * E, device=dev)
c11, c21 = torch.randn(4 * E, device=dev), torch.randn(4 * E, device=dev)
pos = (torch.arange(T, device=dev, dtype=torch.int32) % 500).contiguous()
ang = torch.outer(torch.arange(500.), 1.0 / (10000 ** (torch.arange(0, d, 2) / d)))
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos().to(torch.bfloat16).to(dev), ang.sin().to(torch.bfloat16).to(dev)
rot = (cos, sin, pos, d, 2 * E)
def run_all():
    out = {}
    qkv = torch.empty(T, 3 * E, device=dev, dtype=torch.bfloat16)
    out['qkv +rot+lnf'] = _hip.gemm_fused(x, wqkv, None, out=qkv, rot=rot, ln=(stats, E, 1e-5, c1q, c2q)).clone()
    part = torch.zeros(NB, T, 2, device=dev)
    y = x.clone()
    out['out resid+stats'] = _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=part).clone()
    out['out stats'] = part.clone()
    u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16)
    out['ffn1 gelu+lnf'] = _hip.gemm_fused(x, w1, None, _hip.EPI_GELU, out=u, ln=(stats, E, 1e-5, c11, c21)).clone()
    part2 = torch.zeros(NB, T, 2, device=dev)
    y2 = x.clone()
    out['ffn2 resid+stats'] = _hip.gemm_fused(h4, w2, b2, _hip.EPI_RESIDUAL, y2, 1.0, y2, stats_out=part2).clone()
    out['ffn2 stats'] = part2.clone()
    out['qkv plain'] = _hip.gemm_fused(x, wqkv, None, out=qkv).clone()
    return out
res = {}
for p in (0, 1):
    _hip.set_gemm_options(persist=p)
    res[p] = run_all()
    torch.cuda.synchronize()
ok = True
for k in res[0]:
    eq = bool(torch.equal(res[0][k], res[1][k]))
    ok &= eq
    print(f'{k:20s} persistent == per-tile: {eq}   max |diff| {float((res[0][k].float() - res[1][k].float()).abs().max()):.3e}')
if os.environ.get('TIME', '1') == '1':
    y = x.clone(); u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16); qkv = torch.empty(T, 3 * E, device=dev, dtype=torch.bfloat16)
    part = torch.zeros(NB, T, 2, device=dev)
    fns = {'qkv +rot+lnf': lambda: _hip.gemm_fused(x, wqkv, None, out=qkv, rot=rot, ln=(stats, E, 1e-5, c1q, c2q)),
           'out resid+stats': lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=part),
           'ffn1 gelu+lnf': lambda: _hip.gemm_fused(x, w1, None, _hip.EPI_GELU, out=u, ln=(stats, E, 1e-5, c11, c21)),
           'ffn2 resid+stats': lambda: _hip.gemm_fused(h4, w2, b2, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=part)}
    times = {(k, p): [] for k in fns for p in (0, 1)}
    for r in range(4):
        for k, fn in fns.items():
            for p in (0, 1):
                _hip.set_gemm_options(persist=p)
                fn(); fn()
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(20):
                    fn()
                en.record(); torch.cuda.synchronize()
                times[(k, p)].append(st.elapsed_time(en) / 20 * 1e3)
    _hip.set_gemm_options(persist=1)
    for k in fns:
        a, b = statistics.median(times[(k, 0)]), statistics.median(times[(k, 1)])
        print(f'{k:20s} per-tile {a:7.1f} us   persistent {b:7.1f} us   ({100 * (b / a - 1):+.1f} %)')
sys.exit(0 if ok else 1)
