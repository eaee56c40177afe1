#!/usr/bin/env python
"""A/B the four production GEMMs of an ESM2-650M layer (with and without their fused
epilogues), interleaved over several rounds; prints median / min per variant."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get('EPI_ABLATE', '0') == '1':      # the 'no C store' / 'loop only' hooks live in the TRACE=1 build
    os.environ.setdefault('ESME_HIP_LIB', os.path.join(ROOT, 'esm-efficient_amd', 'esme', 'libesme_hip_trace.so'))
sys.path.insert(0, os.path.join(ROOT, 'esm-efficient_amd'))
import torch
from esme import _hip

T, E, H = int(os.environ.get('T', 50000)), 1280, 20
d = E // H
dev = 'cuda'
torch.manual_seed(0)
def bf(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale).to(torch.bfloat16)
x = bf(T, E); h4 = bf(T, 4 * E)
wqkv, bqkv = bf(3 * E, E, scale=E ** -0.5), bf(3 * E, scale=0.1)
wo, bo = bf(E, E, scale=E ** -0.5), bf(E, scale=0.1)
w1, b1 = bf(4 * E, E, scale=E ** -0.5), bf(4 * E, scale=0.1)
w2, b2 = bf(E, 4 * E, scale=(4 * E) ** -0.5), bf(E, scale=0.1)
stats1 = _hip.row_sums(x)
NB = _hip.stats_blocks(T, E)
stats = (stats1 / NB).expand(NB, T, 2).contiguous()      # same sums split over the blocks a residual GEMM emits
c1q, c2q = torch.randn(3 * E, device=dev), torch.randn(3 * E, device=dev)
c11, c21 = torch.randn(4 * E, device=dev), torch.randn(4 * E, device=dev)
partial = torch.empty(_hip.stats_blocks(T, E), T, 2, device=dev)
pos = (torch.arange(T, device=dev, dtype=torch.int32) % 500).contiguous()
ang = torch.outer(torch.arange(500.), 1.0 / (10000 ** (torch.arange(0, d, 2) / d)))
ang = torch.cat((ang, ang), -1)
cos, sin = ang.cos().to(torch.bfloat16).to(dev), ang.sin().to(torch.bfloat16).to(dev)
qkv = torch.empty(T, 3 * E, device=dev, dtype=torch.bfloat16)
u = torch.empty(T, 4 * E, device=dev, dtype=torch.bfloat16)
y = x.clone()
rot = (cos, sin, pos, d, 2 * E)
variants = {
    'qkv plain':            lambda: _hip.gemm_fused(x, wqkv, bqkv, out=qkv),
    'qkv +rot':             lambda: _hip.gemm_fused(x, wqkv, bqkv, out=qkv, rot=rot),
    'qkv +rot+lnf':         lambda: _hip.gemm_fused(x, wqkv, None, out=qkv, rot=rot, ln=(stats, E, 1e-5, c1q, c2q)),
    'qkv +rot+lnf(nblk=1)': lambda: _hip.gemm_fused(x, wqkv, None, out=qkv, rot=rot, ln=(stats1, E, 1e-5, c1q, c2q)),
    'out resid':            lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y),
    'out resid+stats':      lambda: _hip.gemm_fused(x, wo, bo, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=partial),
    'out plain(no resid)':  lambda: _hip.gemm_fused(x, wo, bo, out=y),
    'ffn1 plain':           lambda: _hip.gemm_fused(x, w1, b1, out=u),
    'ffn1 gelu':            lambda: _hip.gemm_fused(x, w1, b1, _hip.EPI_GELU, out=u),
    'ffn1 gelu+lnf':        lambda: _hip.gemm_fused(x, w1, None, _hip.EPI_GELU, out=u, ln=(stats, E, 1e-5, c11, c21)),
    'ffn2 resid':           lambda: _hip.gemm_fused(h4, w2, b2, _hip.EPI_RESIDUAL, y, 1.0, y),
    'ffn2 resid+stats':     lambda: _hip.gemm_fused(h4, w2, b2, _hip.EPI_RESIDUAL, y, 1.0, y, stats_out=partial),
    'ffn2 plain(no resid)': lambda: _hip.gemm_fused(h4, w2, b2, out=y),
}
def dbg(v, fn):
    def run():
        _hip.load().esme_hip_debug_set_gemm_nt(v)
        fn()
        _hip.load().esme_hip_debug_set_gemm_nt(0)
    return run
if os.environ.get('EPI_ABLATE', '0') == '1':
    for k in ('qkv +rot+lnf', 'out resid+stats', 'ffn1 gelu+lnf', 'ffn2 resid+stats'):
        variants[k + ' [no C store]'] = dbg(2, variants[k])
        variants[k + ' [loop only]'] = dbg(3, variants[k])
flops = {'qkv': 2 * T * 3 * E * E, 'out': 2 * T * E * E, 'ffn1': 2 * T * 4 * E * E, 'ffn2': 2 * T * 4 * E * E}
ROUNDS, ITERS = int(os.environ.get('ROUNDS', 5)), int(os.environ.get('ITERS', 20))
times = {k: [] for k in variants}
for fn in variants.values():
    fn()
torch.cuda.synchronize()
for r in range(ROUNDS):
    for name, fn in variants.items():
        fn(); fn()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(ITERS):
            fn()
        en.record(); torch.cuda.synchronize()
        times[name].append(st.elapsed_time(en) / ITERS * 1e3)
for name, ts in times.items():
    fl = flops[name.split()[0]]
    med, mn = statistics.median(ts), min(ts)
    print(f'{name:24s} median {med:7.1f} us ({fl / med / 1e6:7.1f} TF)   min {mn:7.1f} us ({fl / mn / 1e6:7.1f} TF)', flush=True)
